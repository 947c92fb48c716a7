#!/bin/bash
# Builds the C-ABI HIP library for gfx950 in-tree (the .so travels to the GPU box with gpurun).
set -e
cd "$(dirname "$0")"
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Iinclude \
  -save-temps=obj -Rpass-analysis=kernel-resource-usage \
  -o build/libtai_sepconv.so video-frame-inpainting_amd/csrc/sepconv_capi.hip 2> build/resource_usage.txt \
  || { cat build/resource_usage.txt; exit 1; }
cp build/libtai_sepconv.so video-frame-inpainting_amd/libtai_sepconv.so
grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" build/resource_usage.txt | sed 's/.*remark: [^ ]* *//' | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
