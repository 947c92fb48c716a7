// Where does a chunk of the split-bf16 Winograd kernel (csrc/wino_split.hip.inc) spend its cycles?  Standalone (no torch): launches the
// kernel's DBG build -- wave 0 of every workgroup stamps the shader clock at entry, after the prologue, after every chunk, at the
// end -- in the full form and with parts of the chunk loop left out (SKIP bits: 1 no transform / split arithmetic, 2 no LDS writes
// of the patches, 4 no operand reads, 8 no weight DMA, 16 no patch loads, 32 no barriers in the chunk loop, 64 no MFMAs; results are then wrong,
// only the timing means something).  Per variant: kernel time by HIP events and the median over workgroups of the prologue, the
// median chunk and the epilogue in shader cycles.
// Build: hipcc -O3 -std=c++17 -fno-slp-vectorize -w --offload-arch=gfx950 tools/wino_split_ablate.hip -o build/wino_split_ablate
// Run:   ./build/wino_split_ablate [N C K H W]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../video-frame-inpainting_amd/csrc/wino_conv.hip.inc"
#include "../video-frame-inpainting_amd/csrc/wino_split.hip.inc"
#include "experiments/r04_split_producer_consumer.hip.inc"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

static void magic(long long d, unsigned& m, unsigned& sh) {
    if (d <= 1) { m = 0; sh = 0; return; }
    int lg = 0;
    while ((2LL << lg) <= d) ++lg;
    if ((1LL << lg) == d) --lg;
    sh = (unsigned)lg;
    const unsigned __int128 num = (unsigned __int128)1 << (32 + lg);
    m = (unsigned)((num + (unsigned __int128)d - 1) / (unsigned __int128)d);
}

struct Problem {
    int N, C, K, H, W, Kpad, Cpad, kblocks, nchunks;
    long long tblocks;
    float *x, *w, *bias, *y;
    unsigned short* U3;
    long long* stamps;
    wino::DivMagic dv;
};

static std::vector<float> g_ref;
static bool g_pc = false;       // argv[6] == "pc": the producer / consumer form (12 waves)

template <bool EDGE, int SKIP, bool PC = false>
static void run(const Problem& p, const char* what) {
    if (g_pc && !PC) { run<EDGE, SKIP, true>(p, what); return; }
    auto kern = PC ? wino::split::conv3x3_pc<1, 0, 0, EDGE, 1, SKIP> : wino::split::conv3x3<1, 0, 0, EDGE, 1, SKIP>;
    const int threads = PC ? 768 : 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, wino::split::LDS_BYTES));
    const unsigned grid = (unsigned)(p.tblocks * p.kblocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), wino::split::LDS_BYTES, 0, p.x, p.x, p.x, p.x, p.C, p.U3, p.bias, p.y, (float*)nullptr,
                           p.N, p.C, p.K, p.H, p.W, p.H, p.W, 0, 0, p.nchunks, p.kblocks, p.H / 2, p.W / 2, 0, 0, (const float*)nullptr,
                           (float*)nullptr, p.dv, p.stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    std::vector<long long> st((size_t)grid * 64);
    CK(hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<long long> pro, epi, chunk, total;
    const int nst = std::min(p.nchunks, 26);
    for (unsigned g = 0; g < grid; ++g) {
        const long long* s = &st[(size_t)g * 64];
        pro.push_back(s[1] - s[0]); epi.push_back(s[3] - s[2]); total.push_back(s[3] - s[0]);
        for (int c = 1; c < nst; ++c) chunk.push_back(s[4 + c] - s[4 + c - 1]);
    }
    auto med = [](std::vector<long long>& v) { if (v.empty()) return 0LL; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    if (SKIP == 0) {          // the two forms must agree bit for bit: keep the first full run's output, compare the later ones
        std::vector<float>& ref = g_ref;
        std::vector<float> out((size_t)p.N * p.K * p.H * p.W);
        CK(hipMemcpy(out.data(), p.y, out.size() * 4, hipMemcpyDeviceToHost));
        if (ref.empty()) ref = out;
        else {
            size_t bad = 0; double worst = 0;
            for (size_t i = 0; i < out.size(); ++i) if (out[i] != ref[i]) { ++bad; worst = std::max(worst, (double)fabsf(out[i] - ref[i])); }
            printf("  %s form against the first full run: %zu of %zu outputs differ (max %.3g)\n", PC ? "producer / consumer" : "two-team", bad, out.size(), worst);
        }
    }
    printf("  SKIP %3d %-44s %8.1f us   prologue %6lld  chunk %6lld  epilogue %6lld  workgroup %7lld cycles\n", SKIP, what, best * 1e3f,
           med(pro), med(chunk), med(epi), med(total));
    fflush(stdout);
}

template <bool EDGE>
static void run_all(const Problem& p) {
    run<EDGE, 0>(p, "full kernel");
    run<EDGE, 1024>(p, "teams in phase");
    run<EDGE, 1>(p, "no transform / split arithmetic");
    run<EDGE, 4>(p, "no operand reads");
    run<EDGE, 8>(p, "no weight DMA");
    run<EDGE, 16>(p, "no patch loads");
    run<EDGE, 8 | 16>(p, "no weight DMA, no patch loads");
    run<EDGE, 32>(p, "no barriers in the chunk loop");
    run<EDGE, 128>(p, "no barrier in front of position 3");
    run<EDGE, 256>(p, "no barrier in front of position 7");
    run<EDGE, 512>(p, "barriers without their lgkmcnt(0)");
    run<EDGE, 64>(p, "no MFMAs");
    run<EDGE, 1 | 2 | 8 | 16>(p, "MFMAs + operand reads + barriers only");
    run<EDGE, 1 | 2 | 4 | 8 | 16 | 32>(p, "MFMAs only");
}

int main(int argc, char** argv) {
    Problem p;
    p.N = 64; p.C = 256; p.K = 128; p.H = 64; p.W = 64;
    if (argc >= 6) { p.N = atoi(argv[1]); p.C = atoi(argv[2]); p.K = atoi(argv[3]); p.H = atoi(argv[4]); p.W = atoi(argv[5]); }
    p.Kpad = (p.K + 63) / 64 * 64; p.Cpad = (p.C + 7) / 8 * 8; p.kblocks = p.Kpad / 64; p.nchunks = p.Cpad / 8;
    const long long tiles = (long long)p.N * (p.H / 2) * (p.W / 2);
    p.tblocks = (tiles + 63) / 64;
    const int tw = p.W / 2;
    if (!(tw % 16 == 0 || (tw >= 2 && (tw & (tw - 1)) == 0))) { printf("tile rows must be 2^k or 16 m tiles\n"); return 1; }
    magic((long long)(p.H / 2) * (p.W / 2), p.dv.m_tpi, p.dv.s_tpi);
    magic(p.W / 2, p.dv.m_tw, p.dv.s_tw);
    magic(p.kblocks, p.dv.m_kb, p.dv.s_kb);
    const size_t nx = (size_t)p.N * p.C * p.H * p.W, nw = (size_t)p.K * p.C * 9, ny = (size_t)p.N * p.K * p.H * p.W;
    std::vector<float> hx(nx), hw(nw), hb(p.K);
    unsigned r = 12345u;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.05f;
    for (auto& v : hb) v = rnd();
    CK(hipMalloc(&p.x, nx * 4)); CK(hipMalloc(&p.w, nw * 4)); CK(hipMalloc(&p.bias, p.K * 4)); CK(hipMalloc(&p.y, ny * 4));
    CK(hipMalloc(&p.U3, (size_t)p.Kpad * p.Cpad * 48 * 2)); CK(hipMalloc(&p.stamps, (size_t)p.tblocks * p.kblocks * 64 * 8));
    CK(hipMemcpy(p.x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(p.w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.bias, hb.data(), p.K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(p.stamps, 0, (size_t)p.tblocks * p.kblocks * 64 * 8));
    hipLaunchKernelGGL(wino::split::transform_weights, dim3(1024), dim3(256), 0, 0, p.w, p.U3, p.K, p.C, p.Kpad, p.Cpad);
    CK(hipDeviceSynchronize());
    const bool edge = tw > 16;
    printf("x(%d,%d,%d,%d) -> %d: %lld workgroups, %d chunks, EDGE %d\n", p.N, p.C, p.H, p.W, p.K, p.tblocks * p.kblocks, p.nchunks, (int)edge);
    if (argc >= 7 && !strcmp(argv[6], "pc")) {
        if (edge) run<true, 0, false>(p, "two-team form (reference output)"); else run<false, 0, false>(p, "two-team form (reference output)");
        g_pc = true;
    }
    if (edge) run_all<true>(p); else run_all<false>(p);
    return 0;
}
