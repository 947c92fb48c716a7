#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root): kernel trace + stats, then HBM-traffic PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).  Usage: tools/prof_bench.sh <tag>
tag=${1:-r01}
out=gpurun_out/prof_bench_$tag
mkdir -p $out
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline"
echo "[prof] kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ARGS > $out/trace.log 2>&1
echo "[prof] FETCH_SIZE"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $ARGS > $out/pmc_fetch.log 2>&1
echo "[prof] WRITE_SIZE"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $ARGS > $out/pmc_write.log 2>&1
python3 tools/prof_bench_summary.py $out
