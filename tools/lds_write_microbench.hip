// What does an LDS write cost a wave that is otherwise issuing fp32 MFMAs (1 wave per SIMD, 4 per CU, as in the Winograd
// kernel)?  Per iteration: 8 MFMAs, then NW writes of the given width (lane stride = width: conflict-free), optionally
// NR ds_read_b128 as well.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int NW, int WIDTH, int NR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f16v c0 = {0};
    float a = threadIdx.x, b = threadIdx.x * 0.5f;
    f4v v = {a, b, a + 1, b + 1}, r = {0, 0, 0, 0};
    f2v v2 = {a, b};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = lds + wave * 4096;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if (WIDTH == 16) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lane * 16 + wave * 16384), "v"(v), "n"(q * 1024) : "memory");
            if (WIDTH == 8) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(lane * 8 + wave * 16384), "v"(v2), "n"(q * 512) : "memory");
            if (WIDTH == 4) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(lane * 4 + wave * 16384), "v"(v.x), "n"(q * 256) : "memory");
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lane * 16 + wave * 16384), "n"(q * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = r.x + base[lane];
    for (int j = 0; j < 16; ++j) s += c0[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NW, int WIDTH, int NR> void run(float* out) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 5000; float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(s); hipLaunchKernelGGL((k<NW, WIDTH, NR>), dim3(256), dim3(256), 0, 0, out, iters); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    printf("8 MFMAs + %d x ds_write_b%d + %d x ds_read_b128: %.1f ns per iteration (8 MFMAs alone ~ 224)\n", NW, WIDTH * 8, NR, best * 1e6 / iters);
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<0, 16, 0>(out); run<1, 16, 0>(out); run<2, 16, 0>(out); run<4, 16, 0>(out);
    run<2, 8, 0>(out); run<4, 8, 0>(out); run<8, 8, 0>(out); run<4, 4, 0>(out); run<16, 4, 0>(out);
    run<0, 16, 4>(out); run<2, 16, 4>(out);
    return 0;
}
