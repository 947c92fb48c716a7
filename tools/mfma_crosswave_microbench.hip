// Does vector-ALU work of ANOTHER wave on the same SIMD hide under v_mfma_f32_32x32x2_f32?  (tools/mfma_shadow_microbench.hip
// answered it for the SAME wave: no -- a VALU instruction between fp32 MFMAs costs its full issue time plus a switch.)
// A 512-thread workgroup puts two waves on every SIMD: waves 0-3 run NM MFMAs per iteration, waves 4-7 run NV filler
// instructions per iteration of one kind.  Timed: MFMA waves alone, filler waves alone, both together.  both ~ max(...) means the
// two pipes run side by side; both ~ sum means the fp32 MFMA occupies the vector ALU.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_crosswave_microbench tools/mfma_crosswave_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// KIND: 0 v_add_f32, 1 v_pk_add_f32, 2 v_accvgpr_read_b32, 3 ds_read_b128, 4 bf16 MFMA in the partner (matrix || matrix control),
// 5 v_fma_f32, 6 global store of 16 B per lane
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int run_mfma, int run_fill, int nv) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.f;
    __syncthreads();
    float a = threadIdx.x, b = threadIdx.x * 0.5f;
    if (wave < 4) {
        if (!run_mfma) return;
        f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "v"(b));
        }
        float s = 0;
        for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!run_fill) return;
        float r0 = a, r1 = b, r2 = a + 1, r3 = b + 1, r4 = a + 2, r5 = b + 2, r6 = a + 3, r7 = b + 3;
        v2f p0 = {a, b}, p1 = {b, a}, p2 = {a + 1, b}, p3 = {b, a + 1};
        f16v acc = {0};
        f4v l0 = {0}, l1 = {0};
        const unsigned laddr = (threadIdx.x & 63) * 16;
        for (int i = 0; i < iters; ++i) {
            for (int q = 0; q < nv; q += 8) {
                if (KIND == 0) {
                    asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                                 "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
                } else if (KIND == 5) {
                    asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                                 "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
                } else if (KIND == 1) {
                    asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                                 "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));
                } else if (KIND == 2) {
                    asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %8\n\tv_accvgpr_read_b32 %2, %8\n\tv_accvgpr_read_b32 %3, %8\n\t"
                                 "v_accvgpr_read_b32 %4, %8\n\tv_accvgpr_read_b32 %5, %8\n\tv_accvgpr_read_b32 %6, %8\n\tv_accvgpr_read_b32 %7, %8"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "a"(acc[0]));
                } else if (KIND == 3) {
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\tds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072\n\t"
                                 "ds_read_b128 %0, %2 offset:4096\n\tds_read_b128 %1, %2 offset:5120\n\tds_read_b128 %0, %2 offset:6144\n\tds_read_b128 %1, %2 offset:7168\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(l0), "=&v"(l1) : "v"(laddr) : "memory");
                } else if (KIND == 4) {
                    typedef short s8v __attribute__((ext_vector_type(8)));
                    s8v x = {1, 2, 3, 4, 5, 6, 7, 8};
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, %0"
                                 : "+a"(acc) : "v"(x));
                } else if (KIND == 6) {
                    f4v* dst = reinterpret_cast<f4v*>(out + 1048576) + (size_t)(blockIdx.x * 256 + (threadIdx.x - 256)) * 8;
#pragma unroll
                    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(l0, dst + u);
                }
            }
        }
        float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + l0.x + l1.y + acc[3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int KIND> void run(float* out, const char* what, int nv) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4000;
    float t[3];
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(s);
            hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, iters, mode != 1, mode != 0, nv);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
        }
        t[mode] = best;
    }
    printf("%-22s %3d per 4 MFMAs: mfma alone %.3f ms, filler alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f) -> overlap %.2f\n", what, nv, t[0],
           t[1], t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1], (t[0] + t[1] - t[2]) / (t[0] < t[1] ? t[0] : t[1]));
}
int main() {
    float* out; hipMalloc(&out, (size_t)64 << 20);
    for (int nv : {16, 32, 64}) {
        run<0>(out, "v_add_f32", nv); run<5>(out, "v_fma_f32", nv); run<1>(out, "v_pk_add_f32", nv); run<2>(out, "v_accvgpr_read_b32", nv);
        run<3>(out, "ds_read_b128", nv);
    }
    run<4>(out, "bf16 mfma 32x32x16", 8); run<4>(out, "bf16 mfma 32x32x16", 16);
    run<6>(out, "global store 16 B", 8);
    return 0;
}
