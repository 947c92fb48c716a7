// Sustained fp32 MFMA rate of the whole chip: every wave issues back-to-back v_mfma_f32_32x32x2f32 on 4 independent
// accumulators.  Build: hipcc -w -O3 --offload-arch=gfx950 tools/mfma_microbench.hip -o build/mfma_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int XF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    if (XF == 2) {          // random operands (data-dependent power): a fresh pair of registers every MFMA
        float r[16];
        for (int j = 0; j < 16; ++j) r[j] = out[(threadIdx.x * 16 + j) & 65535];
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[0], r[1], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[2], r[3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[4], r[5], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[6], r[7], c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[8], r[9], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[10], r[11], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[12], r[13], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[14], r[15], c3, 0, 0, 0);
        }
    }
    if (XF == 0) for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    if (XF == 1) {          // one accumulator, every MFMA depends on the one before it
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c0, 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    {
        float* h = (float*)malloc(65536 * 4);
        unsigned st = 12345;
        for (int i = 0; i < 65536; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) / 16777216.0f - 0.5f) * 0.01f; }
        hipMemcpy(out, h, 65536 * 4, hipMemcpyHostToDevice);
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        for (int rep = 0; rep < 4; ++rep) {
            int iters = 20000;
            hipMemcpy(out, h, 65536 * 4, hipMemcpyHostToDevice);
            hipEventRecord(s);
            hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, 0.f, 0.f);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            double flops = (double)256 * 4 * iters * 8 * 32 * 32 * 2 * 2;
            printf("random operands, 1 wave/SIMD: %.2f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
        }
    }
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
        for (int rep = 0; rep < 2; ++rep) {
            int iters = 20000;
            hipEventRecord(s);
            hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, 0.f, 0.f);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            double flops = (double)256 * 4 * iters * 4 * 32 * 32 * 2 * 2;
            printf("dependent chain, 1 wave/SIMD: %.2f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
            hipEventRecord(s);
            hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, 0.f, 0.f);
            hipEventRecord(e); hipEventSynchronize(e);
            hipEventElapsedTime(&ms, s, e);
            printf("4 accumulators,  1 wave/SIMD: %.2f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
