#!/bin/bash
# Collects the round's measurement files on the GPU box into gpurun_out/evidence (copied to profiles/ afterwards).
set -o pipefail
out=gpurun_out/evidence; mkdir -p $out
export TMPDIR=/tmp
echo "[1] tests"; python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest_gpu.txt
echo "[2] bench"; python bench.py 2>$out/bench_stderr.log | tail -1 > $out/bench_line.json; cat $out/bench_line.json | cut -c1-300
echo "[3] wino bench"; timeout -k 10 300 python tools/wino_bench.py > $out/wino_conv_layers.txt 2>&1; tail -3 $out/wino_conv_layers.txt
echo "[4] timeline"; (timeout -k 10 200 python tools/wino_timeline.py; timeout -k 10 200 python tools/wino_timeline.py --skip; timeout -k 10 100 python tools/wino_group_timeline.py) > $out/wino_timeline.txt 2>&1; tail -2 $out/wino_timeline.txt
echo "[5] conv paths"; timeout -k 10 400 python tools/conv_path_times.py > $out/conv_path_times.txt 2>&1; tail -1 $out/conv_path_times.txt
echo "[6] torch profile"; timeout -k 10 300 python tools/torch_profile.py 40 2>/dev/null | grep -v Warn > $out/forward_kernel_table.txt; head -4 $out/forward_kernel_table.txt
echo "[7] microbench"; (./build/mfma_microbench; ./build/mfma_shadow) > $out/mfma_microbench.txt 2>&1; tail -2 $out/mfma_microbench.txt
echo "[8] secondary configs"; timeout -k 10 500 python tools/bench_configs.py 2>/dev/null | grep "^{" > $out/secondary_configs.txt; cat $out/secondary_configs.txt | cut -c1-200
echo "[8b] training step"; timeout -k 10 500 python tools/train_step_bench.py 2>&1 | grep "train\]" > $out/train_step.txt; cat $out/train_step.txt
echo "[9] rocprof bench"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/rocprof_bench.log 2>&1; ls $out/rocprof_bench/*/ | head -5
echo done
