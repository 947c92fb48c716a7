// C ABI of the MI355X-native adaptive separable convolution (see include/tai_sepconv.h).
//
// Replaces the reference's cffi-exported shim SeparableConvolution_cuda_forward / _backward
// (src/separable_convolution/cfile/SeparableConvolution_cuda.c:8-25, :28-51) and its launchers
// (SeparableConvolution_kernel.cu:164-185, :187-242).  gfx950 only; no host synchronisation,
// allocation or copy on any path, so every call can be captured into a hipGraph.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_set>

#include "tai_sepconv.h"

#include "sepconv_fwd.hip.inc"
#include "sepconv_bwd.hip.inc"
#include "upsample.hip.inc"
#include "bias_act.hip.inc"
#include "thin_conv.hip.inc"
#include "wino_conv.hip.inc"
#include "wino_split.hip.inc"
#include "wino43_conv.hip.inc"
#include "wino_wrw.hip.inc"
#include "spectral_norm.hip.inc"
#include "hbm_probe.hip.inc"

namespace {

thread_local char g_err[256] = "";
// Kernel selectors (benchmarking / tests).  Process-wide; relaxed atomics so a selector flipped by one thread while
// another launches is a defined (if unordered) read, never a torn one.
std::atomic<int> g_fwd_variant{0};
std::atomic<int> g_gi_variant{0};   // 0 automatic, 1 force the gather kernel
std::atomic<int> g_vh_variant{0};   // 0 automatic (fused asm kernel for C == 1 when both gradients are wanted), 1 HIP kernels

int fail(int code, const char* fmt, const char* what) {
    std::snprintf(g_err, sizeof(g_err), fmt, what);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return TAI_SEPCONV_ELAUNCH;
    }
    return TAI_SEPCONV_OK;
}

bool dims_ok(int B, int C, int H, int W, int ks) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ks <= 0) return false;
    const long long lim = 0x7fffffffLL;
    const long long Hp = H + ks - 1, Wp = W + ks - 1;
    return (long long)B * C * Hp * Wp < lim && (long long)B * ks * H * W < lim;
}

// Raises the kernel's dynamic-LDS limit past the default 64 KiB.  The attribute is per DEVICE and sticks, so it is set
// once per (device, kernel, size) and remembered only after the runtime accepted it: the steady-state launch path makes
// no runtime call besides hipGetDevice and the launch itself.
template <typename KernelT>
int allow_lds(KernelT kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return TAI_SEPCONV_OK;
    constexpr int SLOTS = 64;
    static thread_local const void* done_kernel[SLOTS];
    static thread_local size_t done_bytes[SLOTS];
    static thread_local int done_device[SLOTS];
    static thread_local int n_done = 0;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return fail(TAI_SEPCONV_ELAUNCH, "%s", "hipGetDevice");
    const void* key = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < n_done; ++i)
        if (done_kernel[i] == key && done_device[i] == device && done_bytes[i] >= bytes) return TAI_SEPCONV_OK;
    const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", bytes, hipGetErrorString(e));
        return TAI_SEPCONV_ELAUNCH;
    }
    if (n_done < SLOTS) { done_kernel[n_done] = key; done_bytes[n_done] = bytes; done_device[n_done] = device; ++n_done; }
    return TAI_SEPCONV_OK;
}

template <int KS, int NC, int SPLIT>
int launch_fwd_tiled(const float* in, const float* v, const float* h, float* out, int B, int C, int c0,
                     int H, int W, hipStream_t s) {
    using K = fwd::Cfg<KS, SPLIT>;
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W;
    const int tiles_y = (H + K::TILE_H - 1) / K::TILE_H;
    const size_t lds = K::lds_bytes(NC);
    auto kern = fwd::sepconv_forward_tiled<KS, NC, SPLIT>;
    if (int rc = allow_lds(kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(K::THREADS), lds, s, in, v, h, out, C, c0,
                       H, W, tiles_x, tiles_y);
    return check_launch("sepconv_forward_tiled");
}

template <int KS, int NC>
int launch_fwd_packed(const float* in, const float* v, const float* h, float* out, int B, int C, int c0,
                      int H, int W, hipStream_t s) {
    using K = fwd::Cfg<KS, 1>;
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W;
    const int tiles_y = (H + K::TILE_H - 1) / K::TILE_H;
    const size_t lds = K::lds_bytes(NC);
    auto kern = fwd::sepconv_forward_packed<KS, NC>;
    if (int rc = allow_lds(kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(K::THREADS), lds, s, in, v, h, out, C, c0,
                       H, W, tiles_x, tiles_y);
    return check_launch("sepconv_forward_packed");
}

template <bool STAGGER, int DBG = 0, int WAVES = 4, int ASMV = 0>
int fwd_asm_all_channels(const float* in, const float* v, const float* h, float* out, int B, int C, int H,
                         int W, hipStream_t s) {
    constexpr int TILE_H = 2 * WAVES;
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W;
    const int tiles_y = (H + TILE_H - 1) / TILE_H;
    const size_t patch = (size_t)(TILE_H + 50) * 180 * sizeof(float);
    const size_t lds = ((patch + 1023) & ~(size_t)1023) + (size_t)WAVES * TAI_FWD_ROWLOOP_RING_SLOTS * 1024;
    auto kern = fwd::sepconv_forward_asm<STAGGER, DBG, WAVES, ASMV>;
    if (int rc = allow_lds(kern, lds)) return rc;
    for (int c0 = 0; c0 < C; ++c0) {
        hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(WAVES * 64), lds, s, in, v, h, out, C, c0, H, W,
                           tiles_x, tiles_y);
        if (int rc = check_launch("sepconv_forward_asm")) return rc;
    }
    return TAI_SEPCONV_OK;
}

template <int MIXMODE, int DBG = 0>
int fwd_ab_all_channels(const float* in, const float* v, const float* h, float* out, int B, int C, int H, int W,
                        hipStream_t s) {
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W;
    const int tiles_y = (H + 15) / 16;
    const size_t patch = (size_t)(16 + 50) * 180 * sizeof(float);
    const size_t lds = ((patch + 1023) & ~(size_t)1023) + (size_t)8 * TAI_FWD_ROWLOOP_RING_SLOTS * 1024 + (MIXMODE == 5 ? 16 : 0);
    auto kern = fwd::sepconv_forward_ab<MIXMODE, DBG>;
    if (int rc = allow_lds(kern, lds)) return rc;
    for (int c0 = 0; c0 < C; ++c0) {
        hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(512), lds, s, in, v, h, out, C, c0, H, W, tiles_x,
                           tiles_y);
        if (int rc = check_launch("sepconv_forward_ab")) return rc;
    }
    return TAI_SEPCONV_OK;
}

// compute units of the current device (cached per device and thread, like allow_lds: no runtime call on the launch path after
// the first)
static int device_cu_count() {
    static thread_local int cached_device = -1, cached_cus = 0;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return 0;
    if (device != cached_device) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
        cached_device = device; cached_cus = n;
    }
    return cached_cus;
}

// kernel 20: one persistent workgroup per CU over the tiles of a single-channel frame batch; falls back to kernel 18 when there
// is at most one tile per CU (nothing to overlap) or the tile count is not a multiple of 8 (the XCD-contiguous tile order)
// POLICY: 0 = by footprint (nt loads and the reversed tile walk when the two tap tensors together exceed the Infinity Cache:
// every tap byte is read once and none of it will be there for anybody else), 1 = default cache policy, forward walk (round 3's
// kernel 20), 2 = nt, forward walk, 3 = nt, reversed walk, 4 = default cache policy, reversed walk.
template <int DBG = 0>
int fwd_persistent(const float* in, const float* v, const float* h, float* out, int B, int C, int H, int W, hipStream_t s, bool force,
                   int policy = 0) {
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W, tiles_y = (H + 15) / 16;
    const int ntiles = B * tiles_x * tiles_y;
    const int cus = device_cu_count();
    int grid = cus > 0 ? (cus / 8) * 8 : 0;
    if (C != 1 || grid < 8 || ntiles % 8 != 0 || (!force && ntiles <= grid) || (long long)B * 51 * H * W * 4 > 0xffffffffLL)
        return fwd_ab_all_channels<5>(in, v, h, out, B, C, H, W, s);
    if (grid > ntiles) grid = ntiles;
    const size_t patch = ((size_t)(16 + 50) * 180 * sizeof(float) + 1023) & ~(size_t)1023;
    const size_t lds = 2 * patch + (size_t)8 * TAI_FWD_ROWLOOP_RING_SLOTS * 1024 + 16;
    if (policy == 0) policy = (2LL * B * 51 * H * W * 4 > (256LL << 20)) ? 6 : 1;
#define TAI_LAUNCH_PERSISTENT(NT, REV, APRIO)                                                                        \
    {                                                                                                               \
        auto kern = fwd::sepconv_forward_persistent<DBG, NT, REV, APRIO>;                                           \
        if (int rc = allow_lds(kern, lds)) return rc;                                                               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, in, v, h, out, H, W, tiles_x, tiles_y, ntiles);     \
    }
    if (policy == 5) TAI_LAUNCH_PERSISTENT(true, true, 0)
    else if (policy == 6) TAI_LAUNCH_PERSISTENT(true, true, 1)
    else if (policy == 7) TAI_LAUNCH_PERSISTENT(true, true, 2)
    else if (policy == 3) TAI_LAUNCH_PERSISTENT(true, true, -1)
    else if (policy == 4) TAI_LAUNCH_PERSISTENT(false, true, -1)
    else if (policy == 2) TAI_LAUNCH_PERSISTENT(true, false, -1)
    else TAI_LAUNCH_PERSISTENT(false, false, -1)
#undef TAI_LAUNCH_PERSISTENT
    return check_launch("sepconv_forward_persistent");
}

template <int WAVES>
int fwd_asm_channel_loop(const float* in, const float* v, const float* h, float* out, int B, int C, int H, int W,
                          hipStream_t s) {
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W;
    const int tiles_y = (H + 2 * WAVES - 1) / (2 * WAVES);
    const size_t patch = (size_t)(2 * WAVES + 50) * 180 * sizeof(float);
    const size_t lds = ((patch + 1023) & ~(size_t)1023) + (size_t)WAVES * TAI_FWD_ROWLOOP_RING_SLOTS * 1024;
    auto kern = fwd::sepconv_forward_asm_channels<WAVES>;
    if (int rc = allow_lds(kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(WAVES * 64), lds, s, in, v, h, out, C, H, W, tiles_x, tiles_y);
    return check_launch("sepconv_forward_asm_channels");
}

// channels in groups of three through the three-patch row loop; what is left over through the per-channel loop
template <bool EARLY, int ABL = 0>
int fwd_asm_three_channels(const float* in, const float* v, const float* h, float* out, int B, int C, int H, int W, hipStream_t s) {
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W, tiles_y = (H + 15) / 16;
    const size_t lds = (size_t)3 * TAI_FWD_ROWLOOP_C3_PATCH_BYTES + (size_t)8 * TAI_FWD_ROWLOOP_C3_RING_SLOTS * 1024;
    auto kern = fwd::sepconv_forward_asm_c3<EARLY, ABL>;
    if (int rc = allow_lds(kern, lds)) return rc;
    int c0 = 0;
    for (; c0 + 3 <= C; c0 += 3) {
        hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(512), lds, s, in, v, h, out, C, c0, H, W, tiles_x, tiles_y);
        if (int rc = check_launch("sepconv_forward_asm_c3")) return rc;
    }
    for (; c0 < C; ++c0) {
        const size_t patch = (size_t)(16 + 50) * 180 * sizeof(float);
        const size_t lds1 = ((patch + 1023) & ~(size_t)1023) + (size_t)8 * TAI_FWD_ROWLOOP_RING_SLOTS * 1024;
        auto k1 = fwd::sepconv_forward_ab<4, 0>;       // (kernel 16: its LDS size has no counter word)
        if (int rc = allow_lds(k1, lds1)) return rc;
        hipLaunchKernelGGL(k1, dim3(B * tiles_x * tiles_y), dim3(512), lds1, s, in, v, h, out, C, c0, H, W, tiles_x, tiles_y);
        if (int rc = check_launch("sepconv_forward_ab")) return rc;
    }
    return TAI_SEPCONV_OK;
}

template <int KS>
int fwd_packed_all_channels(const float* in, const float* v, const float* h, float* out, int B, int C,
                            int H, int W, hipStream_t s) {
    int c0 = 0;
    for (; c0 + 3 <= C; c0 += 3)
        if (int rc = launch_fwd_packed<KS, 3>(in, v, h, out, B, C, c0, H, W, s)) return rc;
    for (; c0 < C; ++c0)
        if (int rc = launch_fwd_packed<KS, 1>(in, v, h, out, B, C, c0, H, W, s)) return rc;
    return TAI_SEPCONV_OK;
}

template <int KS, int SPLIT>
int fwd_tiled_all_channels(const float* in, const float* v, const float* h, float* out, int B, int C,
                           int H, int W, hipStream_t s) {
    int c0 = 0;
    for (; c0 + 3 <= C; c0 += 3)
        if (int rc = launch_fwd_tiled<KS, 3, SPLIT>(in, v, h, out, B, C, c0, H, W, s)) return rc;
    for (; c0 < C; ++c0)
        if (int rc = launch_fwd_tiled<KS, 1, SPLIT>(in, v, h, out, B, C, c0, H, W, s)) return rc;
    return TAI_SEPCONV_OK;
}

template <int KS, int NC>
int launch_grad_vh_tiled(const float* gO, const float* in, const float* v, const float* h, float* gV,
                         float* gH, int B, int H, int W, hipStream_t s) {
    using K = fwd::Cfg<KS, 1>;
    const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W;
    const int tiles_y = (H + K::TILE_H - 1) / K::TILE_H;
    const size_t lds = K::lds_bytes(NC);
    const dim3 grid(B * tiles_x * tiles_y), block(K::THREADS);
    if (gV) {
        auto kern = bwd::sepconv_grad_v_tiled<KS, NC>;
        if (int rc = allow_lds(kern, lds)) return rc;
        hipLaunchKernelGGL(kern, grid, block, lds, s, gO, in, h, gV, H, W, tiles_x, tiles_y);
        if (int rc = check_launch("sepconv_grad_v_tiled")) return rc;
    }
    if (gH) {
        auto kern = bwd::sepconv_grad_h_tiled<KS, NC>;
        if (int rc = allow_lds(kern, lds)) return rc;
        hipLaunchKernelGGL(kern, grid, block, lds, s, gO, in, v, gH, H, W, tiles_x, tiles_y);
        if (int rc = check_launch("sepconv_grad_h_tiled")) return rc;
    }
    return TAI_SEPCONV_OK;
}

}  // namespace

extern "C" {

int tai_sepconv_version(void) { return 500; }     // 0.5.0: F(4x4, 3x3) chunk loop as generated assembly, displaced-read blocks, any C; (0.4.1: persistent forward kernel beyond the Infinity Cache: type-A waves at their partners' priority (0.4.0: nt tap loads + reversed walk; source hash))

const char* tai_sepconv_last_error(void) { return g_err; }

#ifndef TAI_SOURCE_HASH
#define TAI_SOURCE_HASH "unknown"
#endif
// -DTAI_SOURCE_HASH="..." from _native.build().  The marker in front lets the loader read the hash from the FILE, without mapping
// a binary it may be about to refuse (and without dlopen's by-name cache handing back a library that was since rebuilt).
static const char k_source_hash[] = "TAI_SOURCE_HASH=" TAI_SOURCE_HASH;
const char* tai_sepconv_source_hash(void) { return k_source_hash + 16; }

int tai_sepconv_set_forward_variant(int variant) { return g_fwd_variant.exchange(variant, std::memory_order_relaxed); }

int tai_sepconv_set_grad_taps_variant(int variant) { return g_vh_variant.exchange(variant, std::memory_order_relaxed); }

int tai_sepconv_set_grad_input_variant(int variant) { return g_gi_variant.exchange(variant, std::memory_order_relaxed); }

int tai_sepconv_default_forward_variant(int C, int W, int ks) {
    const bool tileable = (ks == 51) && (W % 4 == 0);
    return !tileable ? 1 : (C == 1 ? 20 : 19);
}

long long tai_sepconv_forward_bytes(int B, int C, int H, int W, int ks) {
    const long long Hp = H + ks - 1, Wp = W + ks - 1;
    return 4LL * ((long long)B * C * Hp * Wp + 2LL * B * ks * H * W + (long long)B * C * H * W);
}

long long tai_sepconv_backward_bytes(int B, int C, int H, int W, int ks) {
    const long long Hp = H + ks - 1, Wp = W + ks - 1;
    return 4LL * ((long long)B * C * H * W + 2LL * B * C * Hp * Wp + 4LL * B * ks * H * W);
}

int tai_sepconv_forward(const float* input, const float* vertical, const float* horizontal,
                        float* output, int B, int C, int H, int W, int ks, void* hip_stream) {
    g_err[0] = 0;
    if (!input || !vertical || !horizontal || !output) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (!dims_ok(B, C, H, W, ks)) return fail(TAI_SEPCONV_EINVAL, "%s", "bad dimensions");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);

    int variant = g_fwd_variant.load(std::memory_order_relaxed);
    const bool tileable = (ks == 51) && (W % 4 == 0);
    // default: mixed type-A / type-B hand-scheduled kernel for single-channel frames; three channel patches per tap row otherwise
    if (variant == 0) variant = tai_sepconv_default_forward_variant(C, W, ks);
    if (variant != 1 && !tileable)
        return fail(TAI_SEPCONV_EINVAL, "%s", "tiled forward variants need ks == 51 and W % 4 == 0");
    switch (variant) {
        case 1: {
            const int n = B * C * H * W;
            hipLaunchKernelGGL(fwd::sepconv_forward_generic, dim3((n + 255) / 256), dim3(256), 0, s, input,
                               vertical, horizontal, output, n, C, H, W, ks);
            return check_launch("sepconv_forward_generic");
        }
        case 2: return fwd_tiled_all_channels<51, 1>(input, vertical, horizontal, output, B, C, H, W, s);
        case 3: return fwd_tiled_all_channels<51, 2>(input, vertical, horizontal, output, B, C, H, W, s);
        case 4: return fwd_packed_all_channels<51>(input, vertical, horizontal, output, B, C, H, W, s);
        case 5: return fwd_asm_all_channels<false>(input, vertical, horizontal, output, B, C, H, W, s);
        case 6: return fwd_asm_all_channels<true>(input, vertical, horizontal, output, B, C, H, W, s);
        case 7: return fwd_asm_all_channels<false, 0, 8>(input, vertical, horizontal, output, B, C, H, W, s);
        case 8: return fwd_asm_all_channels<true, 0, 8>(input, vertical, horizontal, output, B, C, H, W, s);
        case 9: return fwd_asm_all_channels<false, 0, 8, 1>(input, vertical, horizontal, output, B, C, H, W, s);
        case 10: return fwd_ab_all_channels<0>(input, vertical, horizontal, output, B, C, H, W, s);
        case 11: return fwd_ab_all_channels<1>(input, vertical, horizontal, output, B, C, H, W, s);
        case 12: return fwd_ab_all_channels<2>(input, vertical, horizontal, output, B, C, H, W, s);
        case 13: return fwd_ab_all_channels<3>(input, vertical, horizontal, output, B, C, H, W, s);
        case 16: return fwd_ab_all_channels<4>(input, vertical, horizontal, output, B, C, H, W, s);
        case 18: return fwd_ab_all_channels<5>(input, vertical, horizontal, output, B, C, H, W, s);
        case 14: return fwd_asm_channel_loop<8>(input, vertical, horizontal, output, B, C, H, W, s);
        case 15: return fwd_asm_channel_loop<4>(input, vertical, horizontal, output, B, C, H, W, s);
        case 17: return fwd_asm_three_channels<false>(input, vertical, horizontal, output, B, C, H, W, s);
        case 19: return fwd_asm_three_channels<true>(input, vertical, horizontal, output, B, C, H, W, s);
        case 20: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, g_fwd_variant.load(std::memory_order_relaxed) == 20);
        case 21: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 1);   // A/B: default cache policy, forward walk
        case 22: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 2);   // A/B: nt tap loads, forward walk
        case 23: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 3);   // A/B: nt tap loads, reversed walk
        case 24: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 4);   // A/B: default cache policy, reversed walk
        case 25: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 5);   // A/B: as 23, type A at constant priority 0
        case 26: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 6);   // A/B: as 23, type A at constant priority 1
        case 27: return fwd_persistent(input, vertical, horizontal, output, B, C, H, W, s, true, 7);   // A/B: as 23, type A at constant priority 2
#ifdef TAI_TIMING_VARIANTS   // timing experiments (wrong results by design): tools/ build only, never in the shipped library
        case 117: return fwd_asm_three_channels<true, 1>(input, vertical, horizontal, output, B, C, H, W, s);   // kernel 19 without the v-ring wait
        case 118: return fwd_asm_three_channels<true, 2>(input, vertical, horizontal, output, B, C, H, W, s);   // kernel 19 without the window waits
        case 108: return fwd_ab_all_channels<3, 3>(input, vertical, horizontal, output, B, C, H, W, s);
        case 109: return fwd_ab_all_channels<4, 3>(input, vertical, horizontal, output, B, C, H, W, s);   // kernel 16 with stamps
        case 110: return fwd_ab_all_channels<5, 3>(input, vertical, horizontal, output, B, C, H, W, s);   // kernel 18 with stamps
        case 120: return fwd_persistent<1>(input, vertical, horizontal, output, B, C, H, W, s, true);    // kernel 20 with stamps
        case 123: return fwd_persistent<1>(input, vertical, horizontal, output, B, C, H, W, s, true, 3);  // 23 (round 4's first scheme) with stamps
        case 125: return fwd_persistent<1>(input, vertical, horizontal, output, B, C, H, W, s, true, 5);  // 25 / 26 / 27 with stamps
        case 126: return fwd_persistent<1>(input, vertical, horizontal, output, B, C, H, W, s, true, 6);
        case 127: return fwd_persistent<1>(input, vertical, horizontal, output, B, C, H, W, s, true, 7);
        case 106: return fwd_ab_all_channels<0, 3>(input, vertical, horizontal, output, B, C, H, W, s);
        case 107: return fwd_ab_all_channels<2, 3>(input, vertical, horizontal, output, B, C, H, W, s);
        case 111: return fwd_asm_all_channels<false, 0, 8, 2>(input, vertical, horizontal, output, B, C, H, W, s);
        case 112: return fwd_asm_all_channels<false, 0, 8, 3>(input, vertical, horizontal, output, B, C, H, W, s);
        case 103: return fwd_asm_all_channels<false, 3, 8>(input, vertical, horizontal, output, B, C, H, W, s);
        case 104: return fwd_asm_all_channels<true, 3, 8>(input, vertical, horizontal, output, B, C, H, W, s);
        case 105: return fwd_asm_all_channels<false, 3, 4>(input, vertical, horizontal, output, B, C, H, W, s);
        case 101: return fwd_asm_all_channels<false, 1>(input, vertical, horizontal, output, B, C, H, W, s);
        case 102: return fwd_asm_all_channels<false, 2>(input, vertical, horizontal, output, B, C, H, W, s);
#endif
        default: return fail(TAI_SEPCONV_EINVAL, "%s", "unknown forward variant (values >= 100 exist only in the tools build, -DTAI_TIMING_VARIANTS)");
    }
}

int tai_hbm_read_probe(const void* buffer, long long bytes, int nt, float* sink, void* hip_stream) {
    g_err[0] = 0;
    if (!buffer || !sink || bytes < (1 << 20)) return fail(TAI_SEPCONV_EINVAL, "%s", "hbm_read_probe: needs a buffer of at least 1 MiB and a sink of 4096 floats");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t n4 = (size_t)bytes / 16;
    if (nt) hipLaunchKernelGGL(probe::stream_read<true>, dim3(4096), dim3(256), 0, s, static_cast<const probe::f4v*>(buffer), sink, n4);
    else hipLaunchKernelGGL(probe::stream_read<false>, dim3(4096), dim3(256), 0, s, static_cast<const probe::f4v*>(buffer), sink, n4);
    return check_launch("hbm_read_probe");
}

int tai_bias_act_inplace(float* x, const float* bias, int N, int C, int HW, int act, void* hip_stream) {
    g_err[0] = 0;
    if (!x || !bias) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return fail(TAI_SEPCONV_EINVAL, "%s", "bad dimensions or activation");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const long long n = (long long)N * C * HW;
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const long long work = vec ? n / 4 : n;
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
#define TAI_LAUNCH_BACT(A)                                                                                        \
    if (vec) hipLaunchKernelGGL(bact::bias_act_vec4<A>, dim3(blocks), dim3(256), 0, s, x, bias, n / 4, HW / 4, C); \
    else hipLaunchKernelGGL(bact::bias_act_scalar<A>, dim3(blocks), dim3(256), 0, s, x, bias, n, HW, C)
    if (act == bact::ACT_RELU) { TAI_LAUNCH_BACT(bact::ACT_RELU); }
    else if (act == bact::ACT_TANH) { TAI_LAUNCH_BACT(bact::ACT_TANH); }
    else { TAI_LAUNCH_BACT(bact::ACT_NONE); }
#undef TAI_LAUNCH_BACT
    return check_launch("bias_act");
}

int tai_conv_cin1_forward(const float* x, const float* weight, const float* bias, float* y, int N, int Co, int H, int W,
                          int k, int act, void* hip_stream) {
    g_err[0] = 0;
    if (!x || !weight || !bias || !y) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || Co <= 0 || H <= 0 || W <= 0 || W % 4 != 0 || (k != 3 && k != 5) || act < 0 || act > 1)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv_cin1: needs W % 4 == 0, k in {3, 5}, act in {0, 1}");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const long long work = (long long)N * H * (W / 4);
    const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
    const int cgroups = (work < 4 * 262144 && Co >= 16) ? 4 : 1;      // few waves: split the output channels over gridDim.y
#define TAI_LAUNCH_CIN1(K, A) hipLaunchKernelGGL((thin::conv_cin1<K, A>), dim3(blocks, cgroups), dim3(256), 0, s, x, weight, bias, y, N, Co, H, W)
    if (k == 3 && act == 0) TAI_LAUNCH_CIN1(3, 0);
    else if (k == 3) TAI_LAUNCH_CIN1(3, 1);
    else if (act == 0) TAI_LAUNCH_CIN1(5, 0);
    else TAI_LAUNCH_CIN1(5, 1);
#undef TAI_LAUNCH_CIN1
    return check_launch("conv_cin1");
}

int tai_conv_cin1_forward_maxpool_window(const float* x, const float* weight, const float* bias, float* y, float* ypool, int N,
                                         int Co, int H, int W, int k, int act, int pool_h, int pool_w, int pool_oy, int pool_ox,
                                         void* hip_stream) {
    g_err[0] = 0;
    if (!x || !weight || !bias || !y || !ypool) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || Co <= 0 || H <= 0 || W <= 0 || W % 4 != 0 || H % 2 != 0 || (k != 3 && k != 5) || act < 0 || act > 1)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv_cin1_maxpool: needs W % 4 == 0, even H, k in {3, 5}, act in {0, 1}");
    if (pool_oy < 0 || pool_ox < 0 || pool_h < H / 2 + pool_oy || pool_w < W / 2 + pool_ox)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv_cin1_maxpool: bad pooled-output window");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const long long work = (long long)N * (H / 2) * (W / 4);
    const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
    const int cgroups = (work < 4 * 262144 && Co >= 16) ? 4 : 1;
#define TAI_LAUNCH_CIN1P(K, A) hipLaunchKernelGGL((thin::conv_cin1_pool<K, A>), dim3(blocks, cgroups), dim3(256), 0, s, x, weight, bias, y, ypool, N, Co, H, W, pool_h, pool_w, pool_oy, pool_ox)
    if (k == 3 && act == 0) TAI_LAUNCH_CIN1P(3, 0);
    else if (k == 3) TAI_LAUNCH_CIN1P(3, 1);
    else if (act == 0) TAI_LAUNCH_CIN1P(5, 0);
    else TAI_LAUNCH_CIN1P(5, 1);
#undef TAI_LAUNCH_CIN1P
    return check_launch("conv_cin1_maxpool");
}

int tai_conv_cin1_forward_maxpool(const float* x, const float* weight, const float* bias, float* y, float* ypool, int N, int Co,
                                  int H, int W, int k, int act, void* hip_stream) {
    return tai_conv_cin1_forward_maxpool_window(x, weight, bias, y, ypool, N, Co, H, W, k, act, H / 2, W / 2, 0, 0, hip_stream);
}

int tai_unpool2x_add(const float* x, const float* res, float* out, long long planes, int h, int w, void* hip_stream) {
    g_err[0] = 0;
    if (!x || !res || !out) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (planes <= 0 || h <= 0 || w <= 0 || w % 2 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "unpool2x_add: needs even w");
    const long long work = planes * 2 * h * (2 * w / 4);
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(bact::unpool2x_add, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), x, res, out, planes, h, w);
    return check_launch("unpool2x_add");
}

int tai_convlstm_gates_forward(const float* gates, const float* c, float* new_c, float* new_h, int N, int F, int HW,
                               float forget_bias, void* hip_stream) {
    g_err[0] = 0;
    if (!gates || !c || !new_c || !new_h) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || F <= 0 || HW <= 0 || HW % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "convlstm_gates: needs HW % 4 == 0");
    const long long work = (long long)N * F * (HW / 4);
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(bact::convlstm_gates, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), gates, c, new_c,
                       new_h, N, F, HW / 4, forget_bias);
    return check_launch("convlstm_gates");
}

int tai_sn_power_iteration(float* weight, float* u, float* scratch, float* sigma_out, int out_rows, int in_cols, int Ip,
                           void* hip_stream) {
    g_err[0] = 0;
    if (!weight || !u || !scratch) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (out_rows <= 0 || in_cols <= 0 || Ip <= 0 || Ip > 64) return fail(TAI_SEPCONV_EINVAL, "%s", "sn_power_iteration: bad shape or Ip");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    float* v_raw = scratch;
    float* t_raw = scratch + in_cols;
    float* sigma = sigma_out;
    const dim3 wt_block(snorm::WT_COLS, snorm::WT_ROWGROUPS);
    const int wt_grid = (in_cols + snorm::WT_COLS - 1) / snorm::WT_COLS;
    for (int it = 0; it < Ip; ++it) {
        // the stored u is used as it is (SNDiscriminator.py:20-22); later rounds consume the unnormalised product t_raw
        hipLaunchKernelGGL(snorm::wt_u, dim3(wt_grid), wt_block, 0, stream, weight, it == 0 ? u : t_raw, v_raw, out_rows, in_cols,
                           it == 0 ? 0 : 1);
        hipLaunchKernelGGL(snorm::w_v, dim3(out_rows), dim3(256), 0, stream, weight, v_raw, t_raw, in_cols);
    }
    const long long n = (long long)out_rows * in_cols;
    const long long want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
    hipLaunchKernelGGL(snorm::finish, dim3(blocks), dim3(256), 0, stream, weight, t_raw, u, sigma, out_rows, n);
    return check_launch("sn_power_iteration");
}

int tai_window_scale_bias_lrelu(float* y, const float* bias, const float* inv_scale, int nw, int B, int C, int HW, float slope,
                                void* hip_stream) {
    g_err[0] = 0;
    if (!y || !bias || !inv_scale) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (nw <= 0 || B <= 0 || C <= 0 || HW <= 0 || HW % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "window_scale_bias_lrelu: needs HW % 4 == 0");
    const long long n4 = (long long)nw * B * C * (HW / 4);
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(snorm::window_scale_bias_lrelu, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), y, bias,
                       inv_scale, n4, C * (HW / 4), HW / 4, C, B, slope);
    return check_launch("window_scale_bias_lrelu");
}

int tai_window_scale_lrelu_backward(const float* grad_y, const float* y, const float* inv_scale, float* grad_z, float* grad_scaled,
                                    int nw, int B, int C, int HW, float slope, void* hip_stream) {
    g_err[0] = 0;
    if (!grad_y || !y || !inv_scale || !grad_z || !grad_scaled) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (nw <= 0 || B <= 0 || C <= 0 || HW <= 0 || HW % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "window_scale_lrelu_backward: needs HW % 4 == 0");
    const long long n4 = (long long)nw * B * C * (HW / 4);
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(snorm::window_scale_lrelu_backward, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), grad_y, y,
                       inv_scale, grad_z, grad_scaled, n4, C * (HW / 4), B, slope);
    return check_launch("window_scale_lrelu_backward");
}

int tai_thin_conv_wrw(const float* big, const float* thin, float* dw, float* dbias, float* workspace, int N, int Cb, int H, int W,
                      int k, void* hip_stream) {
    g_err[0] = 0;
    if (!big || !thin || !workspace || (!dw && !dbias)) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || Cb <= 0 || H <= 0 || W <= 0 || W % 4 != 0 || (k != 3 && k != 5) || (long long)N * Cb > 0x7fffffffLL)
        return fail(TAI_SEPCONV_EINVAL, "%s", "thin_conv_wrw: needs W % 4 == 0 and k in {3, 5}");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (k == 3) hipLaunchKernelGGL(thin::thin_wrw<3>, dim3(N * Cb), dim3(256), 0, stream, big, thin, workspace, N, Cb, H, W);
    else hipLaunchKernelGGL(thin::thin_wrw<5>, dim3(N * Cb), dim3(256), 0, stream, big, thin, workspace, N, Cb, H, W);
    if (int rc = check_launch("thin_conv_wrw")) return rc;
    const int total = Cb * (k * k + 1);
    hipLaunchKernelGGL(thin::thin_wrw_reduce, dim3((total + 255) / 256), dim3(256), 0, stream, workspace, dw, dbias, N, Cb, k * k);
    return check_launch("thin_conv_wrw_reduce");
}

int tai_act_maxpool2x2_forward(const float* z, float* y, float* ypool, long long planes, int H, int W, int relu, void* hip_stream) {
    g_err[0] = 0;
    if (!z || !y || !ypool) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (planes <= 0 || H <= 0 || W <= 0 || H % 2 != 0 || W % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "act_maxpool2x2: needs even H and W % 4 == 0");
    const long long work = planes * (H / 2) * (W / 4);
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(bact::act_pool2x2_forward, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), z, y, ypool, planes,
                       H, W, relu ? 1 : 0);
    return check_launch("act_maxpool2x2_forward");
}

int tai_act_maxpool2x2_backward(const float* grad_y, const float* grad_ypool, const float* y, float* grad_z, long long planes, int H,
                                int W, int relu, void* hip_stream) {
    g_err[0] = 0;
    if (!y || !grad_z) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (planes <= 0 || H <= 0 || W <= 0 || H % 2 != 0 || W % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "act_maxpool2x2: needs even H and W % 4 == 0");
    const long long work = planes * (H / 2) * (W / 4);
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(bact::act_pool2x2_backward, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), grad_y, grad_ypool,
                       y, grad_z, planes, H, W, relu ? 1 : 0);
    return check_launch("act_maxpool2x2_backward");
}

int tai_convlstm_gates_backward(const float* gates, const float* c, const float* new_c, const float* grad_new_c,
                                const float* grad_new_h, float* grad_gates, float* grad_c, int N, int F, int HW, float forget_bias,
                                void* hip_stream) {
    g_err[0] = 0;
    if (!gates || !c || !new_c || !grad_gates || !grad_c || (!grad_new_c && !grad_new_h)) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || F <= 0 || HW <= 0 || HW % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "convlstm_gates: needs HW % 4 == 0");
    const long long work = (long long)N * F * (HW / 4);
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(bact::convlstm_gates_backward, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), gates, c,
                       new_c, grad_new_c, grad_new_h, grad_gates, grad_c, N, F, HW / 4, forget_bias);
    return check_launch("convlstm_gates_backward");
}

int tai_conv_shift_stack(const float* x, float* out, int N, int C, int H, int W, int k, void* hip_stream) {
    g_err[0] = 0;
    if (!x || !out) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || W % 4 != 0 || (k != 5 && k != 7))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv_shift_stack: needs W % 4 == 0 and k in {5, 7}");
    const int S = k == 5 ? 2 : 3;
    const long long work = (long long)N * S * S * C * (H + 2) * ((W + 4) / 4);
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(thin::shift_stack, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), x, out, N, C, H, W, S, k);
    return check_launch("conv_shift_stack");
}

int tai_conv_cout1_3x3_forward(const float* x, const float* weight, const float* bias, float* y, int N, int Ci, int H,
                               int W, int act, void* hip_stream) {
    g_err[0] = 0;
    if (!x || !weight || !bias || !y) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || Ci <= 0 || H <= 0 || W <= 0 || W % 4 != 0 || act < 0 || act > 2)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv_cout1: needs W % 4 == 0, act in {0, 1, 2}");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const long long work = (long long)N * H * (W / 4);
    const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
    if (act == 0) hipLaunchKernelGGL(thin::conv_cout1_3x3<0>, dim3(blocks), dim3(256), 0, s, x, weight, bias, y, N, Ci, H, W);
    else if (act == 1) hipLaunchKernelGGL(thin::conv_cout1_3x3<1>, dim3(blocks), dim3(256), 0, s, x, weight, bias, y, N, Ci, H, W);
    else hipLaunchKernelGGL(thin::conv_cout1_3x3<2>, dim3(blocks), dim3(256), 0, s, x, weight, bias, y, N, Ci, H, W);
    return check_launch("conv_cout1_3x3");
}

int tai_conv_cout1_5x5_forward(const float* x, const float* weight, const float* bias, float* y, int N, int Ci, int H, int W,
                               void* hip_stream) {
    g_err[0] = 0;
    if (!x || !weight || !y) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (N <= 0 || Ci <= 0 || H <= 0 || W <= 0 || W % 4 != 0) return fail(TAI_SEPCONV_EINVAL, "%s", "conv_cout1_5x5: needs W % 4 == 0");
    const long long work = (long long)N * H * (W / 4);
    const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
    hipLaunchKernelGGL(thin::conv_cout1_5x5, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), x, weight, bias, y, N, Ci,
                       H, W);
    return check_launch("conv_cout1_5x5");
}

// Arithmetic of the Winograd GEMMs: 0 = fp32 MFMA (the default; every parity claim), 1 = split bf16 (three terms, six products,
// fp32 accumulation: wino_split.hip.inc), opt-in.  The mode decides what tai_conv3x3_wino_weight_floats / _transform_weights
// produce: in mode 1 the buffer holds the fp32 image FOLLOWED by the split image, and the buffer is remembered, so that the forward
// entry points follow the BUFFER they are handed (a shape the split kernel does not take runs the fp32 kernel on the same buffer)
// and a buffer made in one mode can never be read in the other's layout.
static std::atomic<int> g_wino_arith{0};
static std::mutex g_split_mu;
static std::unordered_set<const void*> g_split_bufs;
int tai_conv3x3_wino_set_arithmetic(int mode) {
    g_err[0] = 0;
    if (mode != 0 && mode != 1) return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_set_arithmetic: 0 (fp32 MFMA) or 1 (split bf16)");
    return g_wino_arith.exchange(mode, std::memory_order_relaxed);
}
int tai_conv3x3_wino_get_arithmetic(void) { return g_wino_arith.load(std::memory_order_relaxed); }
// the caller is about to free (or has freed) a buffer that tai_conv3x3_wino_transform_weights wrote: drop its layout record
int tai_conv3x3_wino_forget_weights(const float* U) {
    std::lock_guard<std::mutex> lk(g_split_mu);
    return (int)g_split_bufs.erase(U);
}

long long tai_conv3x3_wino_weight_floats(int K, int C) {
    if (K <= 0 || C <= 0) return 0;
    const long long Kpad = (K + wino::TM - 1) / wino::TM * wino::TM, Cpad = (C + wino::KC - 1) / wino::KC * wino::KC;
    // (split image: 16 positions x 3 bf16 terms per weight = 24 floats' worth)
    return (g_wino_arith.load(std::memory_order_relaxed) == 1 ? 40 : 16) * Kpad * Cpad;
}

int tai_conv3x3_wino_transform_weights(const float* weight, float* U, int K, int C, void* hip_stream) {
    g_err[0] = 0;
    if (!weight || !U || K <= 0 || C <= 0) return fail(TAI_SEPCONV_EINVAL, "%s", "wino transform_weights: bad argument");
    const int Kpad = (K + wino::TM - 1) / wino::TM * wino::TM, Cpad = (C + wino::KC - 1) / wino::KC * wino::KC;
    const long long total = (long long)Kpad * Cpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const bool split = g_wino_arith.load(std::memory_order_relaxed) == 1;
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        if (split) g_split_bufs.insert(U); else g_split_bufs.erase(U);
    }
    hipLaunchKernelGGL(wino::transform_weights, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), weight, U, K,
                       C, Kpad, Cpad);
    if (split)
        hipLaunchKernelGGL(wino::split::transform_weights, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), weight,
                           reinterpret_cast<unsigned short*>(U + 16 * total), K, C, Kpad, Cpad);
    return check_launch("wino_transform_weights");
}

// ---- Winograd F(4x4, 3x3) on the fp32 MFMA pipe (csrc/wino43_conv.hip.inc): opt-in prototype -------------------------------------
long long tai_conv3x3_wino43_weight_floats(int K, int C) {
    if (K <= 0 || C <= 0) return 0;
    const long long Kpad = (K + wino43::TM - 1) / wino43::TM * wino43::TM;
    const long long Cpad = (C + wino43::KC - 1) / wino43::KC * wino43::KC;       // zero weights for the channels past C
    return 36 * Kpad * Cpad;
}

int tai_conv3x3_wino43_transform_weights(const float* weight, float* U, int K, int C, void* hip_stream) {
    g_err[0] = 0;
    if (!weight || !U || K <= 0 || C <= 0)
        return fail(TAI_SEPCONV_EINVAL, "%s", "wino43 transform_weights: bad argument");
    const int Kpad = (K + wino43::TM - 1) / wino43::TM * wino43::TM;
    const int Cpad = (C + wino43::KC - 1) / wino43::KC * wino43::KC;
    const long long total = (long long)Kpad * Cpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino43::transform_weights, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), weight, U, K, C,
                       Kpad, Cpad);
    return check_launch("wino43_transform_weights");
}

// 0: the kernel (generated chunk loop); 101..112 (tools build only): timing ablations / schedule variants of it.  Round 4's compiler-
// scheduled forms (waves 8 / 4) are gone with their transform constants (profiles/r04_wino43_prototype.txt, r05_wino43_forms.txt keep the A/B).
static std::atomic<int> g_wino43_waves{0};
int tai_conv3x3_wino43_set_waves(int waves) {
#ifdef TAI_TIMING_VARIANTS   // (timing only, wrong results; tools/gen_wino43_asm.py ABLATIONS)
    if (waves >= 101 && waves <= 112) return g_wino43_waves.exchange(waves, std::memory_order_relaxed);
#endif
    if (waves != 0) return -1;
    return g_wino43_waves.exchange(waves, std::memory_order_relaxed);
}

// Workgroup placement of the F(4x4, 3x3) kernels (forward and weight gradient): 1 (default) = aware of the 8 XCDs and their L2s (see
// conv3x3_gen / conv3x3_wrw_gen), 0 = the dispatch order of rounds 4-5.  Same results either way; for A/B timing.
static std::atomic<int> g_wino43_placement{1};
int tai_conv3x3_wino43_set_placement(int xcd_aware) { return g_wino43_placement.exchange(xcd_aware ? 1 : 0, std::memory_order_relaxed); }

static int wino43_forward_impl(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N, int C, int K, int H,
                               int W, int act, void* hip_stream, float* ypool = nullptr, const float* addx = nullptr, float* y2 = nullptr) {
    if (!xs || !xs[0] || !U || !bias || !y || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || nparts < 1 || nparts > 4)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43: bad argument (1 to 4 input parts)");
    // one tensor: any C (the transformed weights of the channels past C are zero and the loads of those channels past the tensor's end
    // return 0; inside it they read the next image's first channels, finite values times zero)
    const bool ragged = nparts == 1 && C % wino43::KC != 0;
    if (H % 4 != 0 || W % 4 != 0 || C % nparts != 0 || (!ragged && (C / nparts) % wino43::KC != 0) || act < 0 || act > 2)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43: needs H and W multiples of 4, the channels of a part a multiple of 4 (any C for one part), act in {0, 1, 2}");
    if ((long long)N * C * H * W >= (1LL << 29) || (long long)N * K * H * W >= (1LL << 29))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43: tensor too large (2^29 elements or more)");
    if ((y2 && !addx) || (addx && (act != 0 || ypool)) || (ypool && act == 2))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43_ex: y2 needs addx; addx needs act 0 and no pooled output; no pooled output with tanh");
    const float* p[4] = {xs[0], xs[0], xs[0], xs[0]};
    for (int i = 0; i < nparts; ++i) {
        if (!xs[i]) return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43: null input part");
        p[i] = xs[i];
    }
    const int Kpad = (K + wino43::TM - 1) / wino43::TM * wino43::TM;
    const int kblocks = Kpad / wino43::TM, nchunks = (C + wino43::KC - 1) / wino43::KC, cpart = C / nparts;
    const long long tiles = (long long)N * (H / 4) * (W / 4);
    const long long tblocks = (tiles + wino43::TN - 1) / wino43::TN;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    wino43::Window plain_win{};
    plain_win.dispatch_order = g_wino43_placement.load(std::memory_order_relaxed) ? 0 : 1;
#define TAI_W43_LAUNCH_GEN(A, E)                                                                                                \
    {                                                                                                                           \
        auto kern = wino43::conv3x3_gen<A, E>;                                                                                  \
        if (int rc = allow_lds(kern, wino43::LDS_BYTES)) return rc;                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)(tblocks * kblocks)), dim3(512), wino43::LDS_BYTES, s, p[0], p[1], p[2], p[3],   \
                           cpart, U, bias, y, N, C, K, H, W, Kpad, nchunks, kblocks, ypool, addx, y2, plain_win);               \
    }
#ifdef TAI_TIMING_VARIANTS
#define TAI_W43_LAUNCH_VAR(V)                                                                                                   \
    {                                                                                                                           \
        auto kern = wino43::conv3x3_gen<1, 0, V>;                                                                               \
        if (int rc = allow_lds(kern, wino43::LDS_BYTES)) return rc;                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)(tblocks * kblocks)), dim3(512), wino43::LDS_BYTES, s, p[0], p[1], p[2], p[3],   \
                           cpart, U, bias, y, N, C, K, H, W, Kpad, nchunks, kblocks, ypool, addx, y2, plain_win);               \
        return check_launch("conv3x3_wino43 (ablation)");                                                                       \
    }
    switch (g_wino43_waves.load(std::memory_order_relaxed)) {
        case 101: TAI_W43_LAUNCH_VAR(1) case 102: TAI_W43_LAUNCH_VAR(2) case 103: TAI_W43_LAUNCH_VAR(3) case 104: TAI_W43_LAUNCH_VAR(4)
        case 105: TAI_W43_LAUNCH_VAR(5) case 106: TAI_W43_LAUNCH_VAR(6) case 107: TAI_W43_LAUNCH_VAR(7) case 108: TAI_W43_LAUNCH_VAR(8)
        case 109: TAI_W43_LAUNCH_VAR(9) case 110: TAI_W43_LAUNCH_VAR(10) case 111: TAI_W43_LAUNCH_VAR(11) case 112: TAI_W43_LAUNCH_VAR(12)
        default: break;
    }
#undef TAI_W43_LAUNCH_VAR
#endif
    if (ypool) {
        if (act == 0) TAI_W43_LAUNCH_GEN(0, 1) else TAI_W43_LAUNCH_GEN(1, 1)
    } else if (addx) {
        if (y2) TAI_W43_LAUNCH_GEN(0, 2) else TAI_W43_LAUNCH_GEN(0, 3)
    } else {
        if (act == 0) TAI_W43_LAUNCH_GEN(0, 0) else if (act == 1) TAI_W43_LAUNCH_GEN(1, 0) else TAI_W43_LAUNCH_GEN(2, 0)
    }
#undef TAI_W43_LAUNCH_GEN
    return check_launch("conv3x3_wino43");
}

int tai_conv3x3_wino43_forward(const float* x, const float* U, const float* bias, float* y, int N, int C, int K, int H, int W, int act,
                               void* hip_stream) {
    g_err[0] = 0;
    const float* xs[1] = {x};
    return wino43_forward_impl(xs, 1, U, bias, y, N, C, K, H, W, act, hip_stream);
}

int tai_conv3x3_wino43_forward_parts(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N, int C, int K,
                                     int H, int W, int act, void* hip_stream) {
    g_err[0] = 0;
    return wino43_forward_impl(xs, nparts, U, bias, y, N, C, K, H, W, act, hip_stream);
}

int tai_conv3x3_wino43_forward_ex(const float* const* xs, int nparts, const float* U, const float* bias, float* y, float* ypool,
                                  const float* addx, float* y2, int N, int C, int K, int H, int W, int act, void* hip_stream) {
    g_err[0] = 0;
    return wino43_forward_impl(xs, nparts, U, bias, y, N, C, K, H, W, act, hip_stream, ypool, addx, y2);
}

int tai_conv3x3_wino43_forward_blocks(const float* x, int shift_k, const float* U, const float* bias, float* y, float* ypool, int pool_h,
                                      int pool_w, int pool_oy, int pool_ox, int N, int C, int K, int H, int W, int in_h, int in_w, int in_oy,
                                      int in_ox, int act, void* hip_stream) {
    g_err[0] = 0;
    const int S = (shift_k + 2) / 3;
    if (!x || !U || !bias || !y || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || shift_k < 4 || shift_k > 9 || C % (S * S) != 0 ||
        (C / (S * S)) % wino43::KC != 0 || H % 4 != 0 || W % 4 != 0 || act < 0 || act > 1)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43_blocks: bad argument (4 <= shift_k <= 9, C = S^2 x a multiple of 4, H and W multiples of 4, act 0 / 1)");
    // every read of every block must lie inside the plane: rows in_oy - 1 ... in_oy + H + 3 (S - 1), columns in_ox - 1 ... in_ox + W + 3 (S - 1)
    if (in_oy < 1 || in_ox < 1 || in_h < in_oy + H + 1 + 3 * (S - 1) || in_w < in_ox + W + 1 + 3 * (S - 1))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43_blocks: the input plane does not carry the halo the displaced reads need");
    const int cin = C / (S * S);
    if ((long long)N * cin * in_h * in_w >= (1LL << 29) || (long long)N * K * H * W >= (1LL << 29))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43_blocks: tensor too large (2^29 elements or more)");
    if (ypool && pool_h > 0 && (pool_w % 2 != 0 || pool_ox % 2 != 0 || pool_oy < 0 || pool_ox < 0 || pool_oy + H / 2 > pool_h || pool_ox + W / 2 > pool_w))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino43_blocks: the pooled-output window must be even in pool_w and pool_ox and lie inside its plane");
    const int Kpad = (K + wino43::TM - 1) / wino43::TM * wino43::TM;
    const int kblocks = Kpad / wino43::TM, nchunks = C / wino43::KC;
    const long long tiles = (long long)N * (H / 4) * (W / 4);
    const long long tblocks = (tiles + wino43::TN - 1) / wino43::TN;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const wino43::Window win{in_h, in_w, in_oy, in_ox, S, ypool ? pool_h : 0, pool_w, pool_oy, pool_ox, 3 * S > shift_k ? 1 : 0,
                             g_wino43_placement.load(std::memory_order_relaxed) ? 0 : 1};
#define TAI_W43_LAUNCH_BLOCKS(A, E)                                                                                             \
    {                                                                                                                           \
        auto kern = wino43::conv3x3_gen<A, E, 0, true>;                                                                         \
        if (int rc = allow_lds(kern, wino43::LDS_BYTES)) return rc;                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)(tblocks * kblocks)), dim3(512), wino43::LDS_BYTES, s, x, x, x, x, cin, U, bias, y, N, C, K, H, W, \
                           Kpad, nchunks, kblocks, ypool, (const float*)nullptr, (float*)nullptr, win);                         \
    }
    if (ypool) {
        if (act == 0) TAI_W43_LAUNCH_BLOCKS(0, 1) else TAI_W43_LAUNCH_BLOCKS(1, 1)
    } else {
        if (act == 0) TAI_W43_LAUNCH_BLOCKS(0, 0) else TAI_W43_LAUNCH_BLOCKS(1, 0)
    }
#undef TAI_W43_LAUNCH_BLOCKS
    return check_launch("conv3x3_wino43_blocks");
}

// Split of the weight-gradient kernel's reduction (the tiles) over workgroups: about one workgroup per CU in total.
struct WrwPlan { int kblocks, cblocks, nchunks, chunks_per_split, splits, pair; };
static bool wrw_plan(int N, int C, int K, int H, int W, WrwPlan& p, int in_h = 0, int in_w = 0) {
    if (N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || H % 2 != 0 || W % 16 != 0) return false;
    if (in_h <= 0) { in_h = H; in_w = W; }
    if ((long long)N * C * in_h * in_w * 4 >= (1LL << 31) || (long long)N * K * H * W * 4 >= (1LL << 31)) return false;
    p.kblocks = (K + 63) / 64;
    p.cblocks = (C + 63) / 64;
    p.nchunks = (int)((long long)N * (H / 2) * (W / 2) / wino::wrw::CT);
    int want = 256 / (p.kblocks * p.cblocks);
    if (want < 1) want = 1;
    if (want > p.nchunks) want = p.nchunks;
    p.chunks_per_split = (p.nchunks + want - 1) / want;
    p.pair = W % 32 == 0;                         // chunk pairs over 16 consecutive tiles: whole 128-byte lines per load
    if (p.pair && (p.chunks_per_split & 1)) ++p.chunks_per_split;      // (the chunk count is even when W % 32 == 0)
    p.splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
    return true;
}

// The same gradient in the F(4x4, 3x3) domain (wino43::conv3x3_wrw_gen): blocks of 64 output x 32 input channels, chunks of four tiles
// (16 pixels of a row), the run of chunks split over about one workgroup per CU; the slabs have the F(2x2) kernel's layout
// [split][tap][Kpad][Cpad] (Cpad a multiple of 64) and go through the same wrw_reduce.
static bool wrw43_plan(int N, int C, int K, int H, int W, WrwPlan& p, int in_h = 0, int in_w = 0, int in_oy = 0, int in_ox = 0) {
    if (N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || H % 4 != 0 || W % 16 != 0) return false;
    if (in_h <= 0) { in_h = H; in_w = W; }
    // an input plane with a halo must hold the whole one-pixel frame of the output window (nothing is padded then)
    if ((in_h != H || in_w != W) && (in_oy < 1 || in_ox < 1 || in_oy + H + 1 > in_h || in_ox + W + 1 > in_w)) return false;
    if ((long long)N * C * in_h * in_w * 4 + (in_w + 1) * 4 >= (1LL << 31) || (long long)N * K * H * W * 4 >= (1LL << 31)) return false;
    p.kblocks = (K + 63) / 64;
    p.cblocks = (C + 31) / 32;
    p.nchunks = (int)((long long)N * (H / 4) * (W / 16));
    // splits: one workgroup per CU where the blocks divide the 256 CUs; otherwise the count (up to 32, at least 16 chunks each) whose
    // last round of workgroups is fullest -- 144 blocks (the 7x7 layer's stack: 36 x 4) as 1 split leave 112 CUs idle for the whole
    // kernel, as 7 splits 4 rounds of 252 run in 0.57 of that time
    const int blocks = p.kblocks * p.cblocks;
    int want = 1;
    double best = 1e30;
    auto rounds_per_split = [&](int sp) { return (double)((blocks * sp + 255) / 256) / sp; };
    for (int sp = 1; sp <= 32 && sp <= p.nchunks && (sp == 1 || p.nchunks / sp >= 16); ++sp) best = rounds_per_split(sp) < best ? rounds_per_split(sp) : best;
    for (int sp = 1; sp <= 32; ++sp)                 // the smallest count within 3 % of the best (every split writes a slab and runs an epilogue)
        if (rounds_per_split(sp) <= 1.03 * best) { want = sp; break; }
    if (blocks * want < 256) {                       // fewer workgroups than CUs in one round: as many splits as fill it
        want = 256 / blocks;
        if (want > p.nchunks) want = p.nchunks;
    }
    p.chunks_per_split = (p.nchunks + want - 1) / want;
    p.splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
    p.pair = 0;
    return true;
}

static std::atomic<int> g_wrw_tile{4};              // 4 (default): the F(4x4, 3x3)-domain kernel where its shape rules allow, 2: F(2x2, 3x3) always
int tai_conv3x3_wino_wrw_set_tile(int tile) {
    if (tile != 2 && tile != 4) return -1;
    return g_wrw_tile.exchange(tile, std::memory_order_relaxed);
}

long long tai_conv3x3_wino_wrw_workspace_floats(int N, int C, int K, int H, int W) {
    WrwPlan p, q;
    if (!wrw_plan(N, C, K, H, W, p)) return -1;
    long long need = (long long)p.splits * 9 * p.kblocks * 64 * p.cblocks * 64 + (long long)p.splits * p.kblocks * 64;     // taps, then bias partials
    if (wrw43_plan(N, C, K, H, W, q)) {              // (either kernel may serve the call: tai_conv3x3_wino_wrw_set_tile)
        const long long n43 = (long long)q.splits * 9 * q.kblocks * 64 * ((C + 63) / 64 * 64) + (long long)q.splits * q.kblocks * 64;
        if (n43 > need) need = n43;
    }
    return need;
}

static std::atomic<int> g_wrw_pair{1};              // 1: paired chunks (whole-line loads) when W % 32 == 0
int tai_conv3x3_wino_wrw_set_paired(int on) { return g_wrw_pair.exchange(on ? 1 : 0, std::memory_order_relaxed); }

static int wino_wrw_impl(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int N, int C, int K, int H,
                         int W, void* hip_stream, long long* stamps, int in_h = 0, int in_w = 0, int in_oy = 0, int in_ox = 0) {
    g_err[0] = 0;
    if (!x || !dy || !dw || !workspace) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    WrwPlan p;
    if (in_h <= 0) { in_h = H; in_w = W; in_oy = in_ox = 0; }
    // the window of every tile must lie inside the plane or in its zero padding on all sides consistently: the origin may
    // not be negative and an input with a halo (origin > 0) must hold the whole 1-pixel frame
    if (in_oy < 0 || in_ox < 0 || in_oy + H > in_h || in_ox + W > in_w)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_wrw: output window outside the input plane");
    if (!wrw_plan(N, C, K, H, W, p, in_h, in_w))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_wrw: needs even H, W % 16 == 0 and tensors below 2 GiB");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    WrwPlan q;
    if (g_wrw_tile.load(std::memory_order_relaxed) == 4 && !stamps && wrw43_plan(N, C, K, H, W, q, in_h, in_w, in_oy, in_ox)) {
        const int Kpad = q.kblocks * 64, Cpad = (C + 63) / 64 * 64;
        float* wsb43 = dbias ? workspace + (long long)q.splits * 9 * Kpad * Cpad : nullptr;
        auto kern = wino43::conv3x3_wrw_gen;
        if (int rc = allow_lds(kern, wino43::WRW_LDS_BYTES)) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)(q.kblocks * q.cblocks * q.splits)), dim3(512), wino43::WRW_LDS_BYTES, stream, x, dy, workspace,
                           wsb43, N, C, K, H, W, q.kblocks, q.cblocks, Kpad, Cpad, q.chunks_per_split, q.nchunks, in_h, in_w, in_oy, in_ox,
                           g_wino43_placement.load(std::memory_order_relaxed) ? 0 : 1);
        if (int rc = check_launch("conv3x3_wino43_wrw")) return rc;
        const long long rows43 = 9LL * K * (Cpad / 64);
        const int blocks43 = (int)(rows43 < 8192 ? (rows43 < q.kblocks ? q.kblocks : rows43) : 8192);
        hipLaunchKernelGGL(wino::wrw::wrw_reduce, dim3(blocks43), dim3(256), 0, stream, workspace, dw, wsb43, dbias, K, C, Kpad, Cpad, q.splits);
        return check_launch("conv3x3_wino43_wrw_reduce");
    }
    const int grid = p.kblocks * p.cblocks * p.splits;
    float* wsb = dbias ? workspace + (long long)p.splits * 9 * p.kblocks * 64 * p.cblocks * 64 : nullptr;
#define TAI_LAUNCH_WRW(D, P)                                                                                                   \
    do {                                                                                                                       \
        auto kern = wino::wrw::conv3x3_wrw<D, P>;                                                                              \
        if (int rc = allow_lds(kern, wino::wrw::LDS_BYTES)) return rc;                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), wino::wrw::LDS_BYTES, stream, x, dy, workspace, wsb, N, C, K, H, W,   \
                           in_h, in_w, in_oy, in_ox, p.kblocks, p.cblocks, p.chunks_per_split, p.nchunks, stamps);                                             \
    } while (0)
    const bool pair = p.pair && g_wrw_pair.load(std::memory_order_relaxed);
    if (stamps) { if (pair) TAI_LAUNCH_WRW(2, true); else TAI_LAUNCH_WRW(2, false); }
    else { if (pair) TAI_LAUNCH_WRW(0, true); else TAI_LAUNCH_WRW(0, false); }
#undef TAI_LAUNCH_WRW
    if (int rc = check_launch("conv3x3_wino_wrw")) return rc;
    const long long rows = 9LL * K * p.cblocks;
    const int blocks = (int)(rows < 8192 ? (rows < p.kblocks ? p.kblocks : rows) : 8192);      // (at least one workgroup per 64 bias entries)
    hipLaunchKernelGGL(wino::wrw::wrw_reduce, dim3(blocks), dim3(256), 0, stream, workspace, dw, wsb, dbias, K, C,
                       p.kblocks * 64, p.cblocks * 64, p.splits);
    return check_launch("conv3x3_wino_wrw_reduce");
}

int tai_conv3x3_wino_wrw(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int N, int C, int K, int H,
                         int W, void* hip_stream) {
    return wino_wrw_impl(x, dy, dw, dbias, workspace, N, C, K, H, W, hip_stream, nullptr);
}

int tai_conv3x3_wino_wrw_window(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int N, int C, int K,
                                int H, int W, int in_h, int in_w, int in_oy, int in_ox, void* hip_stream) {
    if (in_h <= 0 || in_w <= 0) { g_err[0] = 0; return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_wrw_window: bad plane"); }
    return wino_wrw_impl(x, dy, dw, dbias, workspace, N, C, K, H, W, hip_stream, nullptr, in_h, in_w, in_oy, in_ox);
}

#ifdef TAI_TIMING_VARIANTS
int tai_conv3x3_wino_wrw_timeline(const float* x, const float* dy, float* dw, float* workspace, int N, int C, int K, int H,
                                  int W, long long* stamps, void* hip_stream) {
    return wino_wrw_impl(x, dy, dw, nullptr, workspace, N, C, K, H, W, hip_stream, stamps);
}
#endif

struct WinoExtras {                       // optional arguments of the general entry point (tai_conv3x3_wino_forward_ex)
    int shift_s = 0;                      // > 0: ONE input tensor read shift_s x shift_s times, displaced by (3a, 3b)
    int zero_tail = 0;                    // the k x k filter's last block has an all-zero third tap row / column (k % 3 != 0)
    int pool_h = 0, pool_w = 0, pool_oy = 0, pool_ox = 0;     // ypool plane and origin (0: H/2 x W/2 at (0, 0))
    const float* addx = nullptr;          // y2 = y + fixed_unpooling(addx)
    float* y2 = nullptr;
};
static int wino_forward_impl(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N, int C,
                             int K, int H, int W, int act, void* hip_stream, long long* stamps, float* ypool = nullptr,
                             int in_h = 0, int in_w = 0, int in_oy = 0, int in_ox = 0, const WinoExtras& ex = WinoExtras());
static std::atomic<int> g_wino_tall{1};            // 1: use the 128 x 32 workgroup shape when K is a multiple of 128
int tai_conv3x3_wino_set_tall(int on) { return g_wino_tall.exchange(on ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> g_wino_timeline_skip{0};   // timeline launches only: loop parts left out (wino_conv.hip.inc, SKIP)
int tai_conv3x3_wino_timeline_skip(int level) {
#ifdef TAI_TIMING_VARIANTS
    g_wino_timeline_skip.store(level, std::memory_order_relaxed);
    return 0;
#else
    if (level == 0) return 0;
    return fail(TAI_SEPCONV_EINVAL, "%s", "timeline skip levels exist only in the tools build (-DTAI_TIMING_VARIANTS)");
#endif
}

int tai_conv3x3_wino_forward(const float* x, const float* U, const float* bias, float* y, int N, int C, int K, int H, int W,
                             int act, void* hip_stream) {
    const float* xs[4] = {x, x, x, x};
    return wino_forward_impl(xs, 1, U, bias, y, N, C, K, H, W, act, hip_stream, nullptr);
}

int tai_conv3x3_wino_forward_maxpool(const float* x, const float* U, const float* bias, float* y, float* ypool, int N, int C,
                                     int K, int H, int W, int act, void* hip_stream) {
    if (!ypool) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    const float* xs[4] = {x, x, x, x};
    return wino_forward_impl(xs, 1, U, bias, y, N, C, K, H, W, act, hip_stream, nullptr, ypool);
}

int tai_conv3x3_wino_forward_window(const float* x, const float* U, const float* bias, float* y, float* ypool, int N, int C,
                                    int K, int H, int W, int in_h, int in_w, int in_oy, int in_ox, int act, void* hip_stream) {
    const float* xs[4] = {x, x, x, x};
    return wino_forward_impl(xs, 1, U, bias, y, N, C, K, H, W, act, hip_stream, nullptr, ypool, in_h, in_w, in_oy, in_ox);
}

int tai_conv3x3_wino_forward_parts(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N,
                                   int C, int K, int H, int W, int act, void* hip_stream) {
    if (!xs || nparts < 1 || nparts > 4) return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: 1 to 4 input parts");
    if (nparts > 1 && (C % nparts != 0 || (C / nparts) % 8 != 0))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: parts must have equal channel counts, a multiple of 8");
    const float* p[4];
    for (int i = 0; i < 4; ++i) {
        p[i] = xs[i < nparts ? i : 0];
        if (!p[i]) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    }
    return wino_forward_impl(p, nparts, U, bias, y, N, C, K, H, W, act, hip_stream, nullptr);
}

int tai_conv3x3_wino_forward_timeline(const float* x, const float* U, const float* bias, float* y, int N, int C, int K, int H,
                                      int W, long long* stamps, void* hip_stream) {
    if (!stamps) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    const float* xs[4] = {x, x, x, x};
    return wino_forward_impl(xs, 1, U, bias, y, N, C, K, H, W, 1, hip_stream, stamps);
}

#ifdef TAI_TIMING_VARIANTS
static long long* g_wino_ex_stamps = nullptr;
int tai_conv3x3_wino_ex_timeline_target(long long* stamps) { g_wino_ex_stamps = stamps; return 0; }
#endif
int tai_conv3x3_wino_forward_ex(const float* const* xs, int nparts, int shift_k, const float* U, const float* bias, float* y,
                                float* ypool, int pool_h, int pool_w, int pool_oy, int pool_ox, const float* addx, float* y2, int N,
                                int C, int K, int H, int W, int in_h, int in_w, int in_oy, int in_ox, int act, void* hip_stream) {
    if (!xs || nparts < 1 || nparts > 4 || (shift_k != 0 && nparts != 1))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: 1 to 4 input parts, or one tensor read S x S times (shift_k)");
    if (shift_k != 0 && (shift_k < 4 || shift_k > 9))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: shift_k is the size k of the k x k filter, 4 <= k <= 9");
    const int shift_s = shift_k ? (shift_k + 2) / 3 : 0;
    if (nparts > 1 && (C % nparts != 0 || (C / nparts) % 8 != 0))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: parts must have equal channel counts, a multiple of 8");
    if (shift_s != 0 && (C % (shift_s * shift_s) != 0 || (C / (shift_s * shift_s)) % 8 != 0))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: C = S^2 x (a multiple of 8), S = (shift_k + 2) / 3");
    if (y2 && !addx) return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: y2 needs addx");
    const float* p[4];
    for (int i = 0; i < 4; ++i) {
        p[i] = xs[i < nparts ? i : 0];
        if (!p[i]) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    }
    WinoExtras ex;
#ifdef TAI_TIMING_VARIANTS
    if (g_wino_ex_stamps && shift_s) {     // tools build: the next displaced-read launch writes timeline stamps (ReLU kernels only)
        ex.shift_s = shift_s; ex.zero_tail = (3 * shift_s > shift_k) ? 1 : 0;
        long long* st = g_wino_ex_stamps;
        g_wino_ex_stamps = nullptr;
        return wino_forward_impl(p, nparts, U, bias, y, N, C, K, H, W, act, hip_stream, st, nullptr, in_h, in_w, in_oy, in_ox, ex);
    }
#endif
    ex.shift_s = shift_s; ex.zero_tail = (shift_k && 3 * shift_s > shift_k) ? 1 : 0; ex.pool_h = pool_h; ex.pool_w = pool_w; ex.pool_oy = pool_oy; ex.pool_ox = pool_ox; ex.addx = addx; ex.y2 = y2;
    return wino_forward_impl(p, nparts, U, bias, y, N, C, K, H, W, act, hip_stream, nullptr, ypool, in_h, in_w, in_oy, in_ox, ex);
}

static int wino_forward_impl(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N, int C,
                             int K, int H, int W, int act, void* hip_stream, long long* stamps, float* ypool, int in_h,
                             int in_w, int in_oy, int in_ox, const WinoExtras& ex) {
    if (in_h == 0) { in_h = H; in_w = W; }
    g_err[0] = 0;
    if (!xs[0] || !U || !bias || !y) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    const int S = ex.shift_s;
    const int cpart = S ? C / (S * S) : C / nparts;
    const int pool_h = ex.pool_h ? ex.pool_h : H / 2, pool_w = ex.pool_h ? ex.pool_w : W / 2;
    const int pool_oy = ex.pool_h ? ex.pool_oy : 0, pool_ox = ex.pool_h ? ex.pool_ox : 0;
    if (ypool && (pool_oy < 0 || pool_ox < 0 || pool_h < H / 2 + pool_oy || pool_w < W / 2 + pool_ox))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: bad pooled-output window");
    if (ypool && (long long)N * K * pool_h * pool_w >= (1LL << 29))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: pooled tensor too large (2^29 elements or more)");
    // displaced reads stay inside the plane: rows up to H + in_oy + 3 (S - 1), columns up to W + 1 + in_ox + 3 (S - 1)
    if (S && (in_oy < 1 || in_ox < 2 || in_h < H + in_oy + 1 + 3 * (S - 1) || in_w < W + in_ox + 2 + 3 * (S - 1)))
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: the input plane does not hold the halo of the displaced reads");
    if (N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2 || act < 0 || act > 2)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: needs even H and W, act in {0, 1, 2}");
    if (in_h < H + in_oy || in_w < W + in_ox || in_oy < 0 || in_ox < 0 || in_ox % 2 || in_w % 2)
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: bad input window");
    if ((long long)N * C * in_h * in_w >= (1LL << 29) || (long long)N * K * H * W >= (1LL << 29))   // byte offsets < 2^31
        return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino: tensor too large (2^29 elements or more)");
    const int Kpad = (K + wino::TM - 1) / wino::TM * wino::TM, Cpad = (C + wino::KC - 1) / wino::KC * wino::KC;
    const int kblocks = Kpad / wino::TM, nchunks = Cpad / wino::KC;
    const long long tiles = (long long)N * (H / 2) * (W / 2);
    const long long tblocks = (tiles + wino::TN - 1) / wino::TN;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    // A buffer made in split arithmetic (tai_conv3x3_wino_set_arithmetic(1)) takes the split-bf16 kernel where that kernel has the
    // shape: no displaced reads, no timeline stamps, tile rows of 2^k or 16 m tiles (its 16-lane neighbour shifts).
    {
        bool split_buf;
        { std::lock_guard<std::mutex> lk(g_split_mu); split_buf = g_split_bufs.count(U) != 0; }
        const int tw = W / 2;
        const bool tw_ok = tw % 16 == 0 || (tw >= 2 && (tw & (tw - 1)) == 0);
        const int epi_s = ex.addx ? (ex.y2 ? 1 : 2) : 0;
        if (split_buf && !S && !stamps && tw_ok && !(epi_s && act != 0)) {
            const unsigned short* U3 = reinterpret_cast<const unsigned short*>(U + 16LL * Kpad * Cpad);
            const bool edge = tw > 16 || in_ox > 0 || in_w > W + in_ox;
            wino::DivMagic dvS;
            auto magic_s = [](long long d, unsigned& m, unsigned& sh) {
                if (d <= 1) { m = 0; sh = 0; return; }
                int lg = 0;
                while ((2LL << lg) <= d) ++lg;
                if ((1LL << lg) == d) --lg;
                sh = (unsigned)lg;
                const unsigned __int128 num = (unsigned __int128)1 << (32 + lg);
                m = (unsigned)((num + (unsigned __int128)d - 1) / (unsigned __int128)d);
            };
            magic_s((long long)(H / 2) * (W / 2), dvS.m_tpi, dvS.s_tpi);
            magic_s(W / 2, dvS.m_tw, dvS.s_tw);
            magic_s(kblocks, dvS.m_kb, dvS.s_kb);
            const int pm = nparts > 1 ? 1 : 0;
#define TAI_LAUNCH_SPLIT(A, Q, E, G)                                                                                              \
            do {                                                                                                                  \
                auto kern = wino::split::conv3x3<A, Q, E, G>;                                                                     \
                if (int rc = allow_lds(kern, wino::split::LDS_BYTES)) return rc;                                                  \
                hipLaunchKernelGGL(kern, dim3((unsigned)(tblocks * kblocks)), dim3(512), wino::split::LDS_BYTES, s, xs[0], xs[1], \
                                   xs[2], xs[3], cpart, U3, bias, y, ypool, N, C, K, H, W, in_h, in_w, in_oy, in_ox, nchunks,     \
                                   kblocks, pool_h, pool_w, pool_oy, pool_ox, ex.addx, ex.y2, dvS, (long long*)nullptr);          \
            } while (0)
#define TAI_LAUNCH_SPLIT_G(A, Q, E) do { if (edge) TAI_LAUNCH_SPLIT(A, Q, E, true); else TAI_LAUNCH_SPLIT(A, Q, E, false); } while (0)
#define TAI_LAUNCH_SPLIT_Q(A, E) do { if (pm) TAI_LAUNCH_SPLIT_G(A, 1, E); else TAI_LAUNCH_SPLIT_G(A, 0, E); } while (0)
            if (epi_s == 1) TAI_LAUNCH_SPLIT_Q(0, 1);
            else if (epi_s == 2) TAI_LAUNCH_SPLIT_Q(0, 2);
            else if (act == 0) TAI_LAUNCH_SPLIT_Q(0, 0);
            else if (act == 1) TAI_LAUNCH_SPLIT_Q(1, 0);
            else TAI_LAUNCH_SPLIT_Q(2, 0);
#undef TAI_LAUNCH_SPLIT_Q
#undef TAI_LAUNCH_SPLIT_G
#undef TAI_LAUNCH_SPLIT
            return check_launch("conv3x3_wino_split");
        }
    }
    const bool tall = Kpad % wino::TTM == 0 && g_wino_tall.load(std::memory_order_relaxed) != 0;
    const int skip = g_wino_timeline_skip.load(std::memory_order_relaxed);
    (void)skip;
    const int pmode = S ? 2 : (nparts > 1 ? 1 : 0);
#define TAI_WINO_ARGS xs[0], xs[1], xs[2], xs[3], cpart, U, bias, y, ypool, N, C, K, H, W, in_h, in_w, in_oy, in_ox, Kpad, nchunks
    const int part_magic = S ? (1 << 20) / (cpart / 8) + 1 : 0;     // chunk -> channel block of the displaced reads
    // n / d == (n * m) >> (32 + s) for every n < 2^31: s = floor(log2 d), one less for a power of two (m = 2^31, exact); for any
    // other d, 2^s < d gives m = ceil(2^(32+s) / d) < 2^32 and an error term e = m d - 2^(32+s) < d, so n e < 2^(32+s) holds for
    // n <= 2^(32+s) / d, which exceeds 2^31.  d == 1 is flagged by m == 0.
    auto magic = [](long long d, unsigned& m, unsigned& sh) {
        if (d <= 1) { m = 0; sh = 0; return; }
        int lg = 0;
        while ((2LL << lg) <= d) ++lg;                          // floor(log2 d)
        if ((1LL << lg) == d) --lg;
        sh = (unsigned)lg;
        const unsigned __int128 num = (unsigned __int128)1 << (32 + lg);
        m = (unsigned)((num + (unsigned __int128)d - 1) / (unsigned __int128)d);
    };
    wino::DivMagic dvF, dvT;
    magic((long long)(H / 2) * (W / 2), dvF.m_tpi, dvF.s_tpi);
    magic(W / 2, dvF.m_tw, dvF.s_tw);
    magic(kblocks, dvF.m_kb, dvF.s_kb);
    dvT = dvF;
    magic(Kpad / wino::TTM > 0 ? Kpad / wino::TTM : 1, dvT.m_kb, dvT.s_kb);
#define TAI_WINO_TAIL stamps, pool_h, pool_w, pool_oy, pool_ox, ex.addx, ex.y2, S, part_magic, ex.zero_tail
#define TAI_LAUNCH_WINO(A, D, SK, Q, E)                                                                                  \
    do {                                                                                                               \
        if (tall) {      /* 128-channel x 32-tile workgroups (half the patch transform and LDS writes per MFMA) */    \
            const long long ttb = (tiles + wino::TTN - 1) / wino::TTN;                                                 \
            const int tkb = Kpad / wino::TTM;                                                                          \
            if (int rc = allow_lds(wino::conv3x3<A, D, SK, Q, true, E>, wino::TLDS_BYTES)) return rc;                  \
            hipLaunchKernelGGL((wino::conv3x3<A, D, SK, Q, true, E>), dim3((unsigned)(ttb * tkb)), dim3(256), wino::TLDS_BYTES, s, \
                               TAI_WINO_ARGS, tkb, TAI_WINO_TAIL, dvT);                                                \
        } else {                                                                                                       \
            if (int rc = allow_lds(wino::conv3x3<A, D, SK, Q, false, E>, wino::LDS_BYTES)) return rc;                  \
            hipLaunchKernelGGL((wino::conv3x3<A, D, SK, Q, false, E>), dim3((unsigned)(tblocks * kblocks)), dim3(256), wino::LDS_BYTES, s, \
                               TAI_WINO_ARGS, kblocks, TAI_WINO_TAIL, dvF);                                            \
        }                                                                                                              \
    } while (0)
#define TAI_LAUNCH_WINO_ACT(D, SK, Q)                                   \
    do {                                                                \
        if (act == 0) TAI_LAUNCH_WINO(0, D, SK, Q, 0);                  \
        else if (act == 1) TAI_LAUNCH_WINO(1, D, SK, Q, 0);             \
        else TAI_LAUNCH_WINO(2, D, SK, Q, 0);                           \
    } while (0)
    // the instantiations the path uses: the second-output / sum epilogues come without activation (Residual's last
    // convolution) on one tensor or cat operands; the displaced reads come with ReLU (MotionEnc)
    const int epi = ex.addx ? (ex.y2 ? 1 : 2) : 0;
    if (epi && (act != 0 || pmode == 2 || stamps)) return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: addx needs act 0 and no displaced reads");
    if (pmode == 2 && act != 1) return fail(TAI_SEPCONV_EINVAL, "%s", "conv3x3_wino_ex: displaced reads are built for act 1 (ReLU)");
    if (stamps) {        // timeline launches (tools/wino_timeline.py): ReLU, one tensor
#ifdef TAI_TIMING_VARIANTS
        if (skip == 1) TAI_LAUNCH_WINO(1, 1, 1, 0, 0);
        else if (skip == 2) TAI_LAUNCH_WINO(1, 1, 2, 0, 0);
        else if (skip == 4) TAI_LAUNCH_WINO(1, 1, 4, 0, 0);
        else if (skip == 5) TAI_LAUNCH_WINO(1, 1, 5, 0, 0);
        else if (skip == 7) TAI_LAUNCH_WINO(1, 2, 0, 0, 0);
        else
#endif
#ifdef TAI_TIMING_VARIANTS
        if (pmode == 2) TAI_LAUNCH_WINO(1, 1, 0, 2, 0);
        else
#endif
        TAI_LAUNCH_WINO(1, 1, 0, 0, 0);
    }
    else if (epi == 1 && pmode == 1) TAI_LAUNCH_WINO(0, 0, 0, 1, 1);
    else if (epi == 1) TAI_LAUNCH_WINO(0, 0, 0, 0, 1);
    else if (epi == 2 && pmode == 1) TAI_LAUNCH_WINO(0, 0, 0, 1, 2);
    else if (epi == 2) TAI_LAUNCH_WINO(0, 0, 0, 0, 2);
    else if (pmode == 2) TAI_LAUNCH_WINO(1, 0, 0, 2, 0);
    else if (pmode == 1) TAI_LAUNCH_WINO_ACT(0, 0, 1);
    else TAI_LAUNCH_WINO_ACT(0, 0, 0);
#undef TAI_LAUNCH_WINO_ACT
#undef TAI_LAUNCH_WINO
#undef TAI_WINO_TAIL
#undef TAI_WINO_ARGS
    return check_launch("conv3x3_wino");
}

int tai_upsample_bilinear2x_backward(const float* grad_output, float* grad_input, int planes, int H, int W, void* hip_stream) {
    g_err[0] = 0;
    if (!grad_output || !grad_input) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (planes <= 0 || H <= 0 || W <= 0) return fail(TAI_SEPCONV_EINVAL, "%s", "bad dimensions");
    const float rh = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
    const float rw = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const long long work = (long long)planes * H * ((W + 3) / 4);
    const int blocks = (int)((work + 255) / 256 < 65536 ? (work + 255) / 256 : 65536);
    hipLaunchKernelGGL(ups::upsample2x_align_corners_backward, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       grad_output, grad_input, planes, H, W, rh, rw);
    return check_launch("upsample_bilinear2x_backward");
}

int tai_upsample_bilinear2x_forward(const float* input, float* output, int planes, int H, int W, void* hip_stream) {
    g_err[0] = 0;
    if (!input || !output) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (planes <= 0 || H <= 0 || W <= 0) return fail(TAI_SEPCONV_EINVAL, "%s", "bad dimensions");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    // ATen's area_pixel_compute_scale for align_corners = true, in fp32
    const float rh = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
    const float rw = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const long long total = (long long)planes * 2 * H * (2 * W);
    if ((2 * W) % 4 == 0 && H >= 2 && W >= 2 && (long long)H * W * 4 < (1LL << 30)) {
        // two output rows x four columns per thread from a 3 x 4 source window
        const int per_plane = H * (2 * W / 4);
        const int bx = (per_plane + 255) / 256 < 64 ? (per_plane + 255) / 256 : 64;
        hipLaunchKernelGGL(ups::upsample2x_align_corners_pairs, dim3(bx, planes < 65535 ? planes : 65535), dim3(256), 0, s, input, output, planes, H, W,
                           rh, rw);
    } else if ((2 * W) % 4 == 0) {
        const long long threads = total / 4;
        const int blocks = (int)((threads + 255) / 256 < 65536 ? (threads + 255) / 256 : 65536);
        hipLaunchKernelGGL(ups::upsample2x_align_corners_quads, dim3(blocks), dim3(256), 0, s, input, output, planes, H,
                           W, rh, rw);
    } else {
        const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        hipLaunchKernelGGL(ups::upsample2x_align_corners_scalar, dim3(blocks), dim3(256), 0, s, input, output, planes,
                           H, W, rh, rw);
    }
    return check_launch("upsample2x_align_corners");
}

int tai_sepconv_backward(const float* grad_output, const float* input, const float* vertical,
                         const float* horizontal, float* grad_input, float* grad_vertical,
                         float* grad_horizontal, int B, int C, int H, int W, int ks, void* hip_stream) {
    g_err[0] = 0;
    if (!grad_output || !input || !vertical || !horizontal) return fail(TAI_SEPCONV_EINVAL, "%s", "null pointer");
    if (!dims_ok(B, C, H, W, ks)) return fail(TAI_SEPCONV_EINVAL, "%s", "bad dimensions");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);

    const bool tileable = (ks == 51) && (W % 4 == 0) && (C == 1 || C == 3);
    const int gi_variant = g_gi_variant.load(std::memory_order_relaxed);
    bool gi_done = false;
    if (grad_input && tileable && (gi_variant == 0 || gi_variant == 3 || gi_variant == 4)) {
        // gI FIRST (the reference launches V, H, I -- SeparableConvolution_kernel.cu:201-239 -- but the three are
        // independent): wave-private accumulation strips; the tile slabs go to the caller's grad_vertical (or
        // grad_horizontal) buffer, which is filled only afterwards, and a second kernel sums them in a fixed order.
        const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W, tiles_y = (H + bwd::gi2::R - 1) / bwd::gi2::R;
        const long long slab_bytes = (long long)B * tiles_x * tiles_y * C * bwd::gi2::SLAB * (long long)sizeof(float);
        float* scratch = grad_vertical ? grad_vertical : grad_horizontal;
        if (slab_bytes > (long long)B * ks * H * W * (long long)sizeof(float)) scratch = nullptr;   // cannot happen for ks = 51
        const size_t lds = (size_t)bwd::gi2::LDS_FLOATS * sizeof(float);
        const dim3 grid(B * tiles_x * tiles_y), block(512);
        float* dst = scratch ? scratch : grad_input;
        if (!scratch) {      // no buffer to borrow: float atomics on a zeroed gI (last bits then depend on arrival order)
            const size_t bytes = (size_t)B * C * (H + ks - 1) * (W + ks - 1) * sizeof(float);
            if (hipMemsetAsync(grad_input, 0, bytes, s) != hipSuccess) return fail(TAI_SEPCONV_ELAUNCH, "%s", "hipMemsetAsync(gI)");
        }
        const bool use_asm = gi_variant != 4;                  // 4: the HIP C++ row loop (A/B)
        const int to_scratch = scratch ? 1 : 0;
#define TAI_LAUNCH_GI(KERN)                                                                                                  \
    do {                                                                                                                   \
        auto kern = KERN;                                                                                                  \
        if (int rc = allow_lds(kern, lds)) return rc;                                                                      \
        hipLaunchKernelGGL(kern, grid, block, lds, s, grad_output, vertical, horizontal, dst, H, W, tiles_x, tiles_y, to_scratch); \
    } while (0)
        if (C == 1 && use_asm) TAI_LAUNCH_GI(bwd::sepconv_grad_i_strips_asm<1>);
        else if (C == 1) TAI_LAUNCH_GI(bwd::sepconv_grad_i_strips<1>);
        else if (use_asm) TAI_LAUNCH_GI(bwd::sepconv_grad_i_strips_asm<3>);
        else TAI_LAUNCH_GI(bwd::sepconv_grad_i_strips<3>);
#undef TAI_LAUNCH_GI
        if (int rc = check_launch("sepconv_grad_i_strips")) return rc;
        if (scratch) {
            const int n = B * C * (H + ks - 1) * ((W + ks - 1) / 2);
            hipLaunchKernelGGL(bwd::sepconv_grad_i_reduce, dim3((n + 255) / 256), dim3(256), 0, s, scratch, grad_input, n, C, H, W,
                               tiles_x, tiles_y, use_asm ? 2 : 0);
            if (int rc = check_launch("sepconv_grad_i_reduce")) return rc;
        }
        gi_done = true;
    }

    if (tileable && C == 1 && grad_vertical && grad_horizontal && g_vh_variant.load(std::memory_order_relaxed) != 1) {
        // both tap gradients of a single-channel frame in one launch of the hand-scheduled wave types
        const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W, tiles_y = (H + 7) / 8;
        const size_t patch = (size_t)(8 + 50) * 180 * sizeof(float);
        const size_t lds = ((patch + 1023) & ~(size_t)1023) + (size_t)8 * TAI_FWD_ROWLOOP_RING_SLOTS * 1024;
        // (variant 2: the round-2 form that stages the patch behind a workgroup barrier before the tap loads; A/B and tests)
        if (g_vh_variant.load(std::memory_order_relaxed) == 2) {
            auto kern = bwd::sepconv_grad_vh_ab<false>;
            if (int rc = allow_lds(kern, lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(512), lds, s, grad_output, input, vertical, horizontal,
                               grad_vertical, grad_horizontal, H, W, tiles_x, tiles_y);
        } else if (g_vh_variant.load(std::memory_order_relaxed) == 3) {       // A/B: gV waves left at priority 0 (the default until round 4)
            auto kern = bwd::sepconv_grad_vh_ab<true, 0>;
            if (int rc = allow_lds(kern, lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(512), lds, s, grad_output, input, vertical, horizontal,
                               grad_vertical, grad_horizontal, H, W, tiles_x, tiles_y);
        } else if (g_vh_variant.load(std::memory_order_relaxed) == 4) {       // A/B: gV waves at priority 2
            auto kern = bwd::sepconv_grad_vh_ab<true, 2>;
            if (int rc = allow_lds(kern, lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(512), lds, s, grad_output, input, vertical, horizontal,
                               grad_vertical, grad_horizontal, H, W, tiles_x, tiles_y);
        } else {        // gV waves at their gH partners' priority: 112-114 -> 109.6 us at [32,1,128,128], 490 -> 487 at [160,...] (same process)
            auto kern = bwd::sepconv_grad_vh_ab<true, 1>;
            if (int rc = allow_lds(kern, lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(512), lds, s, grad_output, input, vertical, horizontal,
                               grad_vertical, grad_horizontal, H, W, tiles_x, tiles_y);
        }
        if (int rc = check_launch("sepconv_grad_vh_ab")) return rc;
    } else if (tileable) {
        const int rc = (C == 1) ? launch_grad_vh_tiled<51, 1>(grad_output, input, vertical, horizontal,
                                                                grad_vertical, grad_horizontal, B, H, W, s)
                                : launch_grad_vh_tiled<51, 3>(grad_output, input, vertical, horizontal,
                                                                grad_vertical, grad_horizontal, B, H, W, s);
        if (rc) return rc;
    } else {
        const int n = B * ks * H * W;
        if (grad_vertical) {
            hipLaunchKernelGGL(bwd::sepconv_grad_v_generic, dim3((n + 255) / 256), dim3(256), 0, s,
                               grad_output, input, horizontal, grad_vertical, n, C, H, W, ks);
            if (int rc = check_launch("sepconv_grad_v_generic")) return rc;
        }
        if (grad_horizontal) {
            hipLaunchKernelGGL(bwd::sepconv_grad_h_generic, dim3((n + 255) / 256), dim3(256), 0, s,
                               grad_output, input, vertical, grad_horizontal, n, C, H, W, ks);
            if (int rc = check_launch("sepconv_grad_h_generic")) return rc;
        }
    }
    if (grad_input && !gi_done) {
        if (tileable && gi_variant == 2) {
            // first form: LDS row-scatter with a barrier per tap row; accumulates into gI with atomics, so zero it first
            const size_t bytes = (size_t)B * C * (H + ks - 1) * (W + ks - 1) * sizeof(float);
            if (hipMemsetAsync(grad_input, 0, bytes, s) != hipSuccess) return fail(TAI_SEPCONV_ELAUNCH, "%s", "hipMemsetAsync(gI)");
            const int tiles_x = (W + fwd::TILE_W - 1) / fwd::TILE_W, tiles_y = (H + 7) / 8;
            const dim3 grid(B * tiles_x * tiles_y), block(512);
            const size_t lds = ((size_t)C * 58 * 180 + 8 * 2 * 320) * sizeof(float);
            if (C == 1) {
                auto kern = bwd::sepconv_grad_i_rows<51, 1>;
                if (int rc = allow_lds(kern, lds)) return rc;
                hipLaunchKernelGGL(kern, grid, block, lds, s, grad_output, vertical, horizontal, grad_input, H, W, tiles_x, tiles_y);
            } else {
                auto kern = bwd::sepconv_grad_i_rows<51, 3>;
                if (int rc = allow_lds(kern, lds)) return rc;
                hipLaunchKernelGGL(kern, grid, block, lds, s, grad_output, vertical, horizontal, grad_input, H, W, tiles_x, tiles_y);
            }
            if (int rc = check_launch("sepconv_grad_i_rows")) return rc;
            return TAI_SEPCONV_OK;
        }
        const int n = B * (H + ks - 1) * (W + ks - 1);
        const dim3 grid((n + 255) / 256), block(256);
        int c0 = 0;
        for (; c0 + 3 <= C; c0 += 3) {
            hipLaunchKernelGGL(bwd::sepconv_grad_i_gather<3>, grid, block, 0, s, grad_output, vertical,
                               horizontal, grad_input, n, C, c0, H, W, ks);
            if (int rc = check_launch("sepconv_grad_i_gather<3>")) return rc;
        }
        for (; c0 < C; ++c0) {
            hipLaunchKernelGGL(bwd::sepconv_grad_i_gather<1>, grid, block, 0, s, grad_output, vertical,
                               horizontal, grad_input, n, C, c0, H, W, ks);
            if (int rc = check_launch("sepconv_grad_i_gather<1>")) return rc;
        }
    }
    return TAI_SEPCONV_OK;
}

}  // extern "C"
