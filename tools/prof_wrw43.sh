#!/bin/bash
# rocprofv3 evidence for the F(4x4, 3x3)-domain weight-gradient kernel: kernel trace + PMC passes, each its own run, the program directly
# after `--`.  Usage on the GPU box, from the repo root: tools/prof_wrw43.sh
set -e
out=gpurun_out/prof_wrw43
mkdir -p $out
export TMPDIR=/tmp
P="python3 tools/wrw43_run.py 64,256,256,32,32 64,64,64,128,128 64,128,128,64,64 64,512,1024,16,16"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $P > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- $P > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $out/pmc2 -- $P > $out/pmc2.log 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc3 -- $P > $out/pmc3.log 2>&1 || true
python3 tools/prof_w43_summary.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
