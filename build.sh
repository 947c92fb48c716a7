#!/bin/bash
# Builds the C-ABI HIP library for gfx950 in-tree (the .so travels to the GPU box with gpurun), the tools build of it
# (-DTAI_TIMING_VARIANTS, build/libtai_sepconv_timing.so), the oracle and the standalone harness: __graft_entry__.build().
# RESOURCES=1 additionally prints the per-kernel register / LDS usage of the shipped library.
set -e
cd "$(dirname "$0")"
python3 -c "import __graft_entry__ as g; g.build()"
if [ -n "$RESOURCES" ]; then
  mkdir -p build
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Iinclude \
    -save-temps=obj -Rpass-analysis=kernel-resource-usage \
    -o build/libtai_sepconv_resources.so video-frame-inpainting_amd/csrc/sepconv_capi.hip 2> build/resource_usage.txt \
    || { cat build/resource_usage.txt; exit 1; }
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" build/resource_usage.txt | sed 's/.*remark: [^ ]* *//' | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
fi
