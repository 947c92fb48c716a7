#!/bin/bash
# Profile the Winograd weight-gradient kernel with rocprofv3: kernel trace + PMC passes, each its own run.
# Usage on the GPU box, from the repo root: tools/prof_wrw.sh <tag> [N,C,K,H,W ...]
set -e
tag=${1:-r02}
shift || true
out=gpurun_out/prof_wrw_$tag
mkdir -p $out
export TMPDIR=/tmp
P="python3 tools/wrw_bench.py $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $P > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc1 -- $P > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc2 -- $P > $out/pmc2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- $P > $out/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc4 -- $P > $out/pmc4.log 2>&1 || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc5 -- $P > $out/pmc5.log 2>&1 || true
python3 tools/prof_wrw_summary.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
