// What does this box's HBM sustain for a pure streaming READ, a pure WRITE and a copy?  (The sepconv forward's in-model launch reads
// 1.07 GB of taps once and writes 10 MB.)  1 GiB buffers, 16 bytes per lane, grid-stride, several workgroup counts; HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_read_microbench tools/hbm_read_microbench.hip && ./build/hbm_read_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ a, float* __restrict__ out, size_t n4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
        // four independent 16-byte loads in flight per lane
        const float4 x0 = a[i], x1 = i + 256 < n4 ? a[i + 256] : x0, x2 = i + 512 < n4 ? a[i + 512] : x0, x3 = i + 768 < n4 ? a[i + 768] : x0;
        acc.x += x0.x + x1.x + x2.x + x3.x; acc.y += x0.y + x1.y + x2.y + x3.y;
        acc.z += x0.z + x1.z + x2.z + x3.z; acc.w += x0.w + x1.w + x2.w + x3.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;     // never true for the test data: keeps the loads
}
__global__ __launch_bounds__(256) void write_k(float4* __restrict__ b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}

int main() {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {2048, 4096, 8192, 16384}) {
        float ms;
        for (int which = 0; which < 3; ++which) {
            for (int rep = 0; rep < 13; ++rep) {
                if (rep == 3) CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(read_k, dim3(wgs), dim3(256), 0, 0, a, out, n4);
                else if (which == 1) hipLaunchKernelGGL(write_k, dim3(wgs), dim3(256), 0, 0, b, n4);
                else hipLaunchKernelGGL(copy_k, dim3(wgs), dim3(256), 0, 0, a, b, n4);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 10;
            printf("%5d workgroups  %-5s 1 GiB: %7.1f us = %.2f TB/s%s\n", wgs, which == 0 ? "read" : which == 1 ? "write" : "copy", us,
                   (which == 2 ? 2.0 : 1.0) * bytes / us / 1e6, which == 2 ? " (read + write)" : "");
        }
    }
    return 0;
}
