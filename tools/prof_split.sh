#!/bin/bash
# rocprofv3 evidence for the split-bf16 Winograd kernel next to the fp32 kernel on the same shapes (tools/wino_split_bench.py):
# kernel trace + PMC passes, each its own run, the program directly after `--`.
# Usage on the GPU box, from the repo root: tools/prof_split.sh <tag> [N,C,K,H,W ...]
set -e
tag=${1:-r04}
shift || true
out=gpurun_out/prof_split_$tag
mkdir -p $out
export TMPDIR=/tmp
SHAPES=${*:-"64,64,64,128,128 64,128,128,64,64 64,256,256,32,32 64,512,512,16,16"}
P="python3 tools/wino_split_bench.py $SHAPES"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $P > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- $P > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc2 -- $P > $out/pmc2.log 2>&1 || true
python3 tools/prof_split_summary.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
