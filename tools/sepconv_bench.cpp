// Standalone check + timing of the C-ABI sepconv library on one GPU (no torch).
//   sepconv_bench [B C H W] : verifies every forward variant and the backward against the CPU
//   oracle (oracle/libsepconv_oracle.so, test infrastructure) on a small case, then times the
//   variants at the requested shape with HIP events.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "tai_sepconv.h"

extern "C" {
void sepconv_oracle_forward(const float*, const float*, const float*, float*, int, int, int, int, int);
void sepconv_oracle_forward_f64(const float*, const float*, const float*, float*, int, int, int, int, int);
void sepconv_oracle_backward_f64(const float*, const float*, const float*, const float*, float*, float*,
                                 float*, int, int, int, int, int);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<float> rnd(size_t n, unsigned seed, float scale, bool uniform = false) {
    std::mt19937 g(seed); std::vector<float> v(n);
    std::normal_distribution<float> nd(0.f, scale); std::uniform_real_distribution<float> ud(-1.f, 1.f);
    for (auto& x : v) x = uniform ? ud(g) : nd(g);
    return v;
}
static float* up(const std::vector<float>& h) { float* d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d; }
static double maxrel(const std::vector<float>& a, const std::vector<float>& ref) {
    double m = 0; for (size_t i = 0; i < a.size(); ++i) { double d = std::fabs((double)a[i] - ref[i]) / (1 + std::fabs((double)ref[i])); if (!(d <= m)) m = d; } return m;
}

static int check(int B, int C, int H, int W) {
    const int ks = 51, Hp = H + ks - 1, Wp = W + ks - 1;
    auto in = rnd((size_t)B * C * Hp * Wp, 1, 1.f, true), v = rnd((size_t)B * ks * H * W, 2, 0.1f), h = rnd((size_t)B * ks * H * W, 3, 0.1f), gO = rnd((size_t)B * C * H * W, 4, 1.f);
    std::vector<float> ref((size_t)B * C * H * W), out(ref.size());
    sepconv_oracle_forward_f64(in.data(), v.data(), h.data(), ref.data(), B, C, H, W, ks);
    float *din = up(in), *dv = up(v), *dh = up(h), *dgo = up(gO), *dout; CK(hipMalloc(&dout, out.size() * 4));
    int bad = 0;
    for (int var = 1; var <= 27; ++var) {
        CK(hipMemset(dout, 0xff, out.size() * 4));
        tai_sepconv_set_forward_variant(var);
        int rc = tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        double e = maxrel(out, ref);
        printf("  fwd variant %d [%d,%d,%d,%d] rc=%d max|d|/(1+|ref|)=%.3g %s\n", var, B, C, H, W, rc, e, (rc == 0 && e < 1e-5) ? "OK" : "FAIL");
        if (rc != 0 || !(e < 1e-5)) { bad++; printf("    err: %s\n", tai_sepconv_last_error()); }
    }
    tai_sepconv_set_forward_variant(0);
    std::vector<float> gI(in.size()), gV(v.size()), gH(h.size()), rI(in.size()), rV(v.size()), rH(h.size());
    sepconv_oracle_backward_f64(gO.data(), in.data(), v.data(), h.data(), rI.data(), rV.data(), rH.data(), B, C, H, W, ks);
    float *dgI, *dgV, *dgH; CK(hipMalloc(&dgI, gI.size() * 4)); CK(hipMalloc(&dgV, gV.size() * 4)); CK(hipMalloc(&dgH, gH.size() * 4));
    CK(hipMemset(dgI, 0xff, gI.size() * 4)); CK(hipMemset(dgV, 0xff, gV.size() * 4)); CK(hipMemset(dgH, 0xff, gH.size() * 4));
    int rc = tai_sepconv_backward(dgo, din, dv, dh, dgI, dgV, dgH, B, C, H, W, ks, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(gI.data(), dgI, gI.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gV.data(), dgV, gV.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gH.data(), dgH, gH.size() * 4, hipMemcpyDeviceToHost));
    double eI = maxrel(gI, rI), eV = maxrel(gV, rV), eH = maxrel(gH, rH);
    printf("  bwd [%d,%d,%d,%d] rc=%d gI %.3g gV %.3g gH %.3g %s\n", B, C, H, W, rc, eI, eV, eH, (rc == 0 && eI < 2e-5 && eV < 2e-5 && eH < 2e-5) ? "OK" : "FAIL");
    if (rc != 0 || !(eI < 2e-5 && eV < 2e-5 && eH < 2e-5)) { bad++; printf("    err: %s\n", tai_sepconv_last_error()); }
    hipFree(din); hipFree(dv); hipFree(dh); hipFree(dgo); hipFree(dout); hipFree(dgI); hipFree(dgV); hipFree(dgH);
    return bad;
}

static bool want(int var) {   // TAI_VARIANTS="5,101" restricts the timed variants (profiling runs)
    const char* e = getenv("TAI_VARIANTS");
    if (!e) return true;
    char buf[256]; snprintf(buf, sizeof buf, ",%s,", e);
    char key[32]; snprintf(key, sizeof key, ",%d,", var);
    return strstr(buf, key) != nullptr;
}

static void timeit(int B, int C, int H, int W, int iters) {
    const int ks = 51, Hp = H + ks - 1, Wp = W + ks - 1;
    auto in = rnd((size_t)B * C * Hp * Wp, 1, 1.f, true), v = rnd((size_t)B * ks * H * W, 2, 0.1f), h = rnd((size_t)B * ks * H * W, 3, 0.1f), gO = rnd((size_t)B * C * H * W, 4, 1.f);
    float *din = up(in), *dv = up(v), *dh = up(h), *dgo = up(gO), *dout, *dgI, *dgV, *dgH;
    CK(hipMalloc(&dout, (size_t)B * C * H * W * 4)); CK(hipMalloc(&dgI, in.size() * 4)); CK(hipMalloc(&dgV, v.size() * 4)); CK(hipMalloc(&dgH, h.size() * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fb = (double)tai_sepconv_forward_bytes(B, C, H, W, ks), bb = (double)tai_sepconv_backward_bytes(B, C, H, W, ks);
    for (int var = 1; var <= 27; ++var) {
        if (!want(var)) continue;
        tai_sepconv_set_forward_variant(var);
        const int n = var == 1 ? std::max(2, iters / 10) : iters;
        for (int i = 0; i < 3; ++i) tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr);
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / n;
        printf("  fwd variant %d [%d,%d,%d,%d]: %.1f us/call  %.2f TB/s algorithmic (%.1f%% of 8 TB/s)\n", var, B, C, H, W, us, fb / us / 1e6, fb / us / 1e6 / 8 * 100);
    }
    for (int var : {101, 102, 111, 112, 117, 118}) {   // timing experiments: 101 = no tap loads, 102 = tap loads only, 111/112 = wait ablations (C = 1), 117/118 = the same of kernel 19 (C = 3: no v-ring wait / no window waits)
        if (!want(var)) continue;
        tai_sepconv_set_forward_variant(var);
        for (int i = 0; i < 3; ++i) tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr);
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  fwd experiment %d: %.1f us/call\n", var, ms * 1e3 / iters);
    }
    tai_sepconv_set_forward_variant(0);
    struct { const char* n; float *gi, *gv, *gh; } parts[] = {{"gV", nullptr, dgV, nullptr}, {"gH", nullptr, nullptr, dgH}, {"gI", dgI, nullptr, nullptr}, {"all", dgI, dgV, dgH}};
    for (auto& p : parts) {
        if (getenv("TAI_VARIANTS")) break;
        const int n = std::max(2, iters / 10);
        tai_sepconv_backward(dgo, din, dv, dh, p.gi, p.gv, p.gh, B, C, H, W, ks, nullptr);
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) tai_sepconv_backward(dgo, din, dv, dh, p.gi, p.gv, p.gh, B, C, H, W, ks, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  bwd %-3s [%d,%d,%d,%d]: %.1f us/call (all-three algorithmic bytes %.1f MB -> %.2f TB/s if this were all)\n", p.n, B, C, H, W, ms * 1e3 / n, bb / 1e6, bb / (ms * 1e3 / n) / 1e6);
    }
    hipFree(din); hipFree(dv); hipFree(dh); hipFree(dgo); hipFree(dout); hipFree(dgI); hipFree(dgV); hipFree(dgH);
}

int main(int argc, char** argv) {
    int bad = 0;
    if (!getenv("TAI_VARIANTS")) {
        printf("== correctness vs CPU oracle (fp64 accumulate)\n");
        bad += check(2, 1, 16, 128);
        bad += check(1, 3, 12, 36);   // ragged tile (H % 8 != 0, W < 128)
        bad += check(1, 2, 8, 132);   // two column tiles, C = 2 (channel passes)
    }
    if (argc >= 5) {
        printf("== timing\n");
        timeit(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc >= 6 ? atoi(argv[5]) : 50);
    } else {
        printf("== timing cfg2 (B=32 gray 128x128)\n"); timeit(32, 1, 128, 128, 50);
        printf("== timing cfg4 (B=16 RGB 256x256)\n"); timeit(16, 3, 256, 256, 20);
    }
    printf(bad ? "FAILED (%d)\n" : "ALL OK\n", bad);
    return bad ? 1 : 0;
}
