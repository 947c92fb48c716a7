// How much VALU / LDS work fits in the shadow of v_mfma_f32_32x32x2_f32 when both come from the SAME wave
// (1 wave per SIMD, the Winograd kernel's situation)?  Each loop iteration: 4 MFMAs, each followed by NV VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV, int KIND, int NACC = 1>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[4096];
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = threadIdx.x, b = threadIdx.x * 0.5f;
    float r0 = a, r1 = b, r2 = a + 1, r3 = b + 1, r4 = a + 2, r5 = b + 2, r6 = a + 3, r7 = b + 3;
    lds[threadIdx.x] = a;
    __syncthreads();
    if (KIND == 4) {        // clustered: 4 MFMAs back to back, then the 4 * NV VALU ops in one run
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int m = 0; m < 4; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < 4 * NV; ++q)
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(q % 8 == 0 ? r0 : q % 8 == 1 ? r1 : q % 8 == 2 ? r2 : q % 8 == 3 ? r3 : q % 8 == 4 ? r4 : q % 8 == 5 ? r5 : q % 8 == 6 ? r6 : r7) : "v"(b));
        }
    } else if (KIND >= 5) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f p0 = {a, b}, p1 = {b, a}, p2 = {a + 1, b}, p3 = {b, a + 1};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int m = 0; m < 4; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < 4 * NV; ++q) {
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q % 4 == 0 ? p0 : q % 4 == 1 ? p1 : q % 4 == 2 ? p2 : p3) : "v"(p0));
                if (KIND == 6) asm volatile("v_mul_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(q % 8 == 0 ? r0 : q % 8 == 1 ? r1 : q % 8 == 2 ? r2 : q % 8 == 3 ? r3 : q % 8 == 4 ? r4 : q % 8 == 5 ? r5 : q % 8 == 6 ? r6 : r7) : "v"(a), "v"(b));
                if (KIND == 7) asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q % 4 == 0 ? p0 : q % 4 == 1 ? p1 : q % 4 == 2 ? p2 : p3) : "v"(q % 4 == 0 ? p1 : q % 4 == 1 ? p2 : q % 4 == 2 ? p3 : p0));
                if (KIND == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(b));
            }
        }
        r0 += p0.x + p1.y + p2.x + p3.y;
    } else
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (NACC == 1 || m == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            else if (m == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            else if (m == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(q % 8 == 0 ? r0 : q % 8 == 1 ? r1 : q % 8 == 2 ? r2 : q % 8 == 3 ? r3 : q % 8 == 4 ? r4 : q % 8 == 5 ? r5 : q % 8 == 6 ? r6 : r7) : "v"(b));
                if (KIND == 1) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(q % 8 == 0 ? r0 : q % 8 == 1 ? r1 : q % 8 == 2 ? r2 : q % 8 == 3 ? r3 : q % 8 == 4 ? r4 : q % 8 == 5 ? r5 : q % 8 == 6 ? r6 : r7) : : "vcc");
                if (KIND == 2) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(q % 8 == 0 ? r0 : q % 8 == 1 ? r1 : q % 8 == 2 ? r2 : q % 8 == 3 ? r3 : q % 8 == 4 ? r4 : q % 8 == 5 ? r5 : q % 8 == 6 ? r6 : r7) : "v"(b));
                if (KIND == 3) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
            }
        }
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int KIND, int NACC = 1> void run(float* out, const char* what) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 5000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL((k<NV, KIND, NACC>), dim3(256), dim3(256), 0, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    printf("[%d acc] %-10s %2d per MFMA: %.3f ms  -> %.0f ns per MFMA (alone: ~27 ns)\n", NACC, what, NV, best, best * 1e6 / (iters * 4.0));
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<0, 0>(out, "v_add_f32"); run<4, 0>(out, "v_add_f32"); run<8, 0>(out, "v_add_f32"); run<12, 0>(out, "v_add_f32"); run<16, 0>(out, "v_add_f32"); run<24, 0>(out, "v_add_f32");
    run<8, 1>(out, "v_cndmask"); run<16, 1>(out, "v_cndmask");
    run<4, 2>(out, "dpp"); run<8, 2>(out, "dpp");
    run<8, 3>(out, "s_add"); run<16, 3>(out, "s_add");
    run<4, 5>(out, "cl pk_add"); run<8, 5>(out, "cl pk_add"); run<4, 6>(out, "cl mul_dpp"); run<8, 6>(out, "cl mul_dpp"); run<4, 7>(out, "cl pk opsel"); run<8, 7>(out, "cl pk opsel"); run<4, 8>(out, "cl dep add"); run<8, 8>(out, "cl dep add");
    run<1, 4>(out, "clustered"); run<2, 4>(out, "clustered"); run<4, 4>(out, "clustered"); run<8, 4>(out, "clustered"); run<16, 4>(out, "clustered");
    run<1, 0>(out, "v_add_f32"); run<2, 0>(out, "v_add_f32");
    run<0, 0, 4>(out, "v_add_f32"); run<4, 0, 4>(out, "v_add_f32"); run<8, 0, 4>(out, "v_add_f32"); run<12, 0, 4>(out, "v_add_f32"); run<16, 0, 4>(out, "v_add_f32"); run<24, 0, 4>(out, "v_add_f32");
    run<8, 1, 4>(out, "v_cndmask"); run<8, 2, 4>(out, "dpp"); run<16, 3, 4>(out, "s_add");
    return 0;
}
