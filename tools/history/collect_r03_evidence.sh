#!/bin/bash
# Round-3 measurement files, collected on the GPU box into gpurun_out/evidence_r03 (summaries are copied to profiles/).
# rocprofv3: kernel trace and PMC counters in SEPARATE runs, the program directly after `--`.
set -o pipefail
out=gpurun_out/evidence_r03; mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
echo "[1] bench kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -- $BENCH > $out/bench_trace.log 2>&1 || exit 1
echo "[2] bench FETCH_SIZE"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/bench_fetch -- $BENCH > $out/bench_fetch.log 2>&1 || exit 1
echo "[3] bench WRITE_SIZE"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/bench_write -- $BENCH > $out/bench_write.log 2>&1 || exit 1
echo "[4] bench SQ counters"; timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/bench_sq -- $BENCH > $out/bench_sq.log 2>&1 || exit 1
echo "[5] bench LDS counters"; timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/bench_lds -- $BENCH > $out/bench_lds.log 2>&1 || exit 1
echo "[6] sepconv backward"; timeout -k 10 500 tools/prof_bwd.sh r03 > $out/prof_bwd.log 2>&1 || exit 1
echo "[7] sepconv forward C=3 (cfg4 shape), kernels 17 and 19"
export TAI_VARIANTS=17,19
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_trace -- ./build/sepconv_bench 16 3 256 256 30 > $out/c3_trace.log 2>&1 || exit 1
export TAI_VARIANTS=19
S="./build/sepconv_bench 16 3 256 256 6"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c3_fetch -- $S > $out/c3_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/c3_write -- $S > $out/c3_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $out/c3_sq -- $S > $out/c3_sq.log 2>&1 || exit 1
unset TAI_VARIANTS
python3 tools/prof_r03_summary.py $out > $out/summary.log 2>&1
tail -40 $out/summary.log
echo done
