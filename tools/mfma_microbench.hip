// Sustained fp32 MFMA rate of the whole chip: every wave issues back-to-back v_mfma_f32_32x32x2f32 on 4 independent
// accumulators.  Build: hipcc -w -O3 --offload-arch=gfx950 tools/mfma_microbench.hip -o build/mfma_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int XF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    for (int wpb : {1, 2}) {
        int blocks = 256 * wpb * 2;
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        for (int rep = 0; rep < 3; ++rep) {
            int iters = 20000;
            hipEventRecord(s);
            hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.f, 0.f);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            double flops = (double)blocks * 4 * iters * 4 * 32 * 32 * 2 * 2;
            printf("blocks=%d (%d waves/SIMD) %.2f ms  %.1f TFLOP/s fp32 MFMA\n", blocks, blocks * 4 / 1024, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
