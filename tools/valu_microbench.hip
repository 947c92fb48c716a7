// Micro-benchmark: issue cost (cycles per wave-instruction, from s_memtime) of the fp32 FMA forms on
// gfx950 at 1/2/4 waves per SIMD, alone and interleaved with ds_read_b128.  Decides how the sepconv
// inner loop is written.  Each test body is 32 independent instructions, repeated in an asm loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// mode 0: v_fma_f32 (VOP3, 3 distinct VGPR sources)   mode 1: v_fmac_f32 (VOP2)
// mode 2: v_pk_fma_f32                                 mode 3: v_pk_fma_f32 + 1 ds_read_b128 per 8
// mode 4: v_fmac_f32 + 1 ds_read_b128 per 16           mode 5: v_pk_fma_f32, B operand shared (same pair)
template <int MODE>
__global__ void issue_cost(unsigned long long* out, int iters) {
    __shared__ float4 lds[1024];
    lds[threadIdx.x & 1023] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    unsigned long long t0, t1;
    const unsigned ldsaddr = (threadIdx.x & 63) * 16;
    asm volatile(
        "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 0.5\n v_mov_b32 v42, 0.25\n v_mov_b32 v43, 0.125\n"
        "v_mov_b32 v44, 1.0\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.25\n v_mov_b32 v47, 0.125\n"
        "s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"
        "s_mov_b32 s20, %2\n"
        "1:\n"
        ".if %c4 == 0\n"
        REP8("v_fma_f32 v0, v40, v44, v0\n v_fma_f32 v1, v41, v45, v1\n v_fma_f32 v2, v42, v46, v2\n v_fma_f32 v3, v43, v47, v3\n")
        ".endif\n"
        ".if %c4 == 1\n"
        REP8("v_fmac_f32 v0, v40, v44\n v_fmac_f32 v1, v41, v45\n v_fmac_f32 v2, v42, v46\n v_fmac_f32 v3, v43, v47\n")
        ".endif\n"
        ".if %c4 == 2\n"
        REP8("v_pk_fma_f32 v[0:1], v[40:41], v[44:45], v[0:1]\n v_pk_fma_f32 v[2:3], v[42:43], v[46:47], v[2:3]\n v_pk_fma_f32 v[4:5], v[40:41], v[46:47], v[4:5]\n v_pk_fma_f32 v[6:7], v[42:43], v[44:45], v[6:7]\n")
        ".endif\n"
        ".if %c4 == 3\n"
        REP4("ds_read_b128 v[48:51], %3\n v_pk_fma_f32 v[0:1], v[40:41], v[44:45], v[0:1]\n v_pk_fma_f32 v[2:3], v[42:43], v[46:47], v[2:3]\n v_pk_fma_f32 v[4:5], v[40:41], v[46:47], v[4:5]\n v_pk_fma_f32 v[6:7], v[42:43], v[44:45], v[6:7]\n"
             "v_pk_fma_f32 v[8:9], v[40:41], v[44:45], v[8:9]\n v_pk_fma_f32 v[10:11], v[42:43], v[46:47], v[10:11]\n v_pk_fma_f32 v[12:13], v[40:41], v[46:47], v[12:13]\n v_pk_fma_f32 v[14:15], v[42:43], v[44:45], v[14:15]\n")
        "s_waitcnt lgkmcnt(0)\n"
        ".endif\n"
        ".if %c4 == 4\n"
        REP4("ds_read_b128 v[48:51], %3\n" REP4("v_fmac_f32 v0, v40, v44\n v_fmac_f32 v1, v41, v45\n v_fmac_f32 v2, v42, v46\n v_fmac_f32 v3, v43, v47\n"))
        "s_waitcnt lgkmcnt(0)\n"
        ".endif\n"
        ".if %c4 == 5\n"
        REP8("v_pk_fma_f32 v[0:1], v[40:41], v[44:45], v[0:1]\n v_pk_fma_f32 v[2:3], v[42:43], v[44:45], v[2:3]\n v_pk_fma_f32 v[4:5], v[46:47], v[44:45], v[4:5]\n v_pk_fma_f32 v[6:7], v[40:41], v[44:45], v[6:7] op_sel:[1,0,0] op_sel_hi:[1,0,1]\n")
        ".endif\n"
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
        "s_memtime %1\n s_waitcnt lgkmcnt(0)\n"
        : "=&s"(t0), "=&s"(t1)
        : "s"(iters), "v"(ldsaddr), "n"(MODE)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "s20", "scc", "memory");
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned long long* d, int instr_per_iter) {
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        dim3 grid(256), block(256 * wps);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(issue_cost<MODE>, grid, block, 0, 0, d, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(issue_cost<MODE>, grid, block, 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 4 * wps);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[h.size() / 2] / ((double)iters * instr_per_iter);
        // per-SIMD cost: wps waves share one SIMD
        printf("%-34s waves/SIMD=%d  %.2f cyc/instr/wave (memtime ticks)  -> %.2f cyc/instr/SIMD   wall %.3f ms  (%.2f GHz-equiv)\n", name, wps, cyc, cyc / wps, ms,
               (double)h[h.size() / 2] / (ms * 1e6));
    }
}

int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
    run<0>("v_fma_f32 (3 vgpr src)", d, 32);
    run<1>("v_fmac_f32", d, 32);
    run<2>("v_pk_fma_f32", d, 32);
    run<5>("v_pk_fma_f32 shared B / op_sel", d, 32);
    run<3>("8 v_pk_fma_f32 + 1 ds_read_b128", d, 36);
    run<4>("16 v_fmac_f32 + 1 ds_read_b128", d, 68);
    return 0;
}
