// Diagnostic: per-wave timeline of the asm forward kernel (DBG = 3 builds write 100 MHz timestamps instead of pixels).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tai_sepconv.h"
int main(int argc, char** argv) {
    const int var = argc > 1 ? atoi(argv[1]) : 103, waves = (var == 105) ? 4 : 8;
    const int RS = (var >= 106 && var <= 110) ? 8 : 5;      // 8-byte records per wave (the A/B kernels also stamp the shader clock)
    const int B = 32, C = 1, H = 128, W = 128, ks = 51, Hp = H + ks - 1, Wp = W + ks - 1;
    float *din, *dv, *dh, *dout;
    hipMalloc(&din, (size_t)B * Hp * Wp * 4); hipMalloc(&dv, (size_t)B * ks * H * W * 4); hipMalloc(&dh, (size_t)B * ks * H * W * 4); hipMalloc(&dout, (size_t)B * H * W * 4);
    {   // random data: zero operands raise the clock and would flatter the timeline
        std::vector<float> tmp((size_t)B * ks * H * W);
        unsigned s = 12345u;
        auto fill = [&](float* d, size_t n, float sc) { for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; tmp[i] = sc * ((int)(s >> 8) / 8388608.0f - 1.0f); } hipMemcpy(d, tmp.data(), n * 4, hipMemcpyHostToDevice); };
        fill(din, (size_t)B * Hp * Wp, 1.f); fill(dv, (size_t)B * ks * H * W, 0.2f); fill(dh, (size_t)B * ks * H * W, 0.2f);
    }
    tai_sepconv_set_forward_variant(var);
    for (int rep = 0; rep < 3; ++rep) { tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr); hipDeviceSynchronize(); }
    const int nblk = B * (H / (2 * waves)), nw = nblk * waves;
    std::vector<unsigned long long> r((size_t)nw * RS);
    hipMemcpy(r.data(), dout, r.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, tend = 0;
    for (int i = 0; i < nw; ++i) { t0 = std::min(t0, r[i * RS]); tend = std::max(tend, r[i * RS + 3]); }
    printf("variant %d: %d waves, kernel span %.2f us (first wave start -> last wave end)\n", var, nw, (tend - t0) / 100.0);
    double s[4] = {0, 0, 0, 0}; 
    std::vector<double> starts, ends;
    for (int i = 0; i < nw; ++i) { for (int k = 0; k < 4; ++k) s[k] += (r[i * RS + k] - t0) / 100.0; starts.push_back((r[i*RS]-t0)/100.0); ends.push_back((r[i*RS+3]-t0)/100.0); }
    printf("mean (us since first start): start %.2f  taps-early/patch done %.2f  taps done %.2f  rowloop done %.2f\n", s[0] / nw, s[1] / nw, s[2] / nw, s[3] / nw);
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    printf("start p50 %.2f p99 %.2f max %.2f | end p1 %.2f p50 %.2f max %.2f\n", starts[nw/2], starts[nw*99/100], starts[nw-1], ends[nw/100], ends[nw/2], ends[nw-1]);
    for (int blk : {0, nblk / 2}) for (int w = 0; w < waves; ++w) {
        const unsigned long long* q = &r[((size_t)blk * waves + w) * RS];
        const unsigned hw = (unsigned)q[4];
        printf("  blk %4d wave %d: start %6.2f  t1 %6.2f  t2 %6.2f  end %6.2f   simd %u wave_slot %u cu %u se %u\n", blk, w, (q[0] - t0) / 100.0, (q[1] - t0) / 100.0,
               (q[2] - t0) / 100.0, (q[3] - t0) / 100.0, (hw >> 4) & 3, hw & 15, (hw >> 8) & 15, (hw >> 13) & 7);
    }
    return 0;
}
