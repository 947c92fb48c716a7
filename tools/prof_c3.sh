#!/bin/bash
# rocprofv3 evidence for the three-channel forward at the cfg4 shape [16,3,256,256]: kernel trace + PMC passes (separate runs).
set -o pipefail
out=gpurun_out/evidence_r02; mkdir -p $out
export TMPDIR=/tmp
export TAI_VARIANTS=${TAI_VARIANTS:-17}
S="./build/sepconv_bench 16 3 256 256 6"
rm -rf $out/c3_trace $out/c3_fetch $out/c3_write $out/c3_sq
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_trace -- ./build/sepconv_bench 16 3 256 256 30 > $out/c3_trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c3_fetch -- $S > $out/c3_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/c3_write -- $S > $out/c3_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $out/c3_sq -- $S > $out/c3_sq.log 2>&1
python3 tools/prof_r02_summary.py $out 2>/dev/null | tail -42
