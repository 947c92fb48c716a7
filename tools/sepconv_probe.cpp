// Debug probe: delta taps v = e_i, h = e_j must give out[y,x] = in[y+i, x+j] exactly.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tai_sepconv.h"
int main(int argc, char** argv) {
    const int var = argc > 1 ? atoi(argv[1]) : 5;
    const int B = 1, C = 1, H = 8, W = 128, ks = 51, Hp = H + ks - 1, Wp = W + ks - 1;
    std::vector<float> in((size_t)Hp * Wp), v((size_t)ks * H * W), h(v.size()), out((size_t)H * W);
    for (int y = 0; y < Hp; ++y) for (int x = 0; x < Wp; ++x) in[(size_t)y * Wp + x] = 1000.f * y + x;
    float *din, *dv, *dh, *dout;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dv, v.size() * 4); hipMalloc(&dh, h.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    tai_sepconv_set_forward_variant(var);
    int is[] = {0, 1, 2, 7, 25, 49, 50}, js[] = {0, 1, 2, 3, 4, 5, 24, 25, 48, 49, 50};
    for (int i : is) for (int j : js) {
        std::fill(v.begin(), v.end(), 0.f); std::fill(h.begin(), h.end(), 0.f);
        for (int p = 0; p < H * W; ++p) { v[(size_t)i * H * W + p] = 1.f; h[(size_t)j * H * W + p] = 1.f; }
        hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(dout, 0xff, out.size() * 4);
        int rc = tai_sepconv_forward(din, dv, dh, dout, B, C, H, W, ks, nullptr);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        int bad[4] = {0, 0, 0, 0}; int first = -1;
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            float e = in[(size_t)(y + i) * Wp + x + j];
            if (out[y * W + x] != e) { bad[x & 3]++; if (first < 0) first = y * W + x; }
        }
        if (bad[0] + bad[1] + bad[2] + bad[3])
            printf("i=%2d j=%2d rc=%d bad per pixel-in-quad: %d %d %d %d  first (y=%d,x=%d): got %.1f want %.1f\n", i, j, rc, bad[0], bad[1], bad[2], bad[3],
                   first / W, first % W, out[first], in[(size_t)(first / W + i) * Wp + first % W + j]);
        else printf("i=%2d j=%2d OK\n", i, j);
    }
    return 0;
}
