#!/bin/bash
# Profile sepconv forward variants on cfg2 with rocprofv3 (kernel trace + PMC passes, each its own run).
# Usage (on the GPU box, from the repo root): TAI_VARIANTS=5,101 tools/prof_fwd.sh <tag>
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B=./build/sepconv_bench
export TAI_VARIANTS=${TAI_VARIANTS:-5}
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B 32 1 128 128 30 > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc1 -- $B 32 1 128 128 6 > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/pmc2 -- $B 32 1 128 128 6 > $out/pmc2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- $B 32 1 128 128 6 > $out/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc4 -- $B 32 1 128 128 6 > $out/pmc4.log 2>&1 || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc5 -- $B 32 1 128 128 6 > $out/pmc5.log 2>&1 || true
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
