/*
 * sepconv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's adaptive separable convolution, the only
 * native component of MichiganCOG/video-frame-inpainting.  The reference ships
 * no CPU implementation (src/separable_convolution/SeparableConvolution.py:48-49
 * raises NotImplementedError), so the four CUDA kernel bodies are the
 * specification that is restated here loop for loop:
 *
 *   sepconv_oracle_forward  <- kernel_SeparableConvolution_updateOutput
 *                              src/separable_convolution/cfile/SeparableConvolution_kernel.cu:19-47
 *   sepconv_oracle_grad_v   <- kernel_SeparableConvolution_updateGradV  .cu:49-86
 *   sepconv_oracle_grad_h   <- kernel_SeparableConvolution_updateGradH  .cu:88-118
 *   sepconv_oracle_grad_i   <- kernel_SeparableConvolution_updateGradI  .cu:120-162
 *   sepconv_oracle_backward <- SeparableConvolution_kernel_backward     .cu:187-242
 *
 * Same loop nesting, same operand order inside each product, same fp32
 * accumulator (`float dblOutput`, .cu:33) as the reference; the *_f64 twins keep
 * the loops but accumulate in double, and are used by the tests to measure how
 * far any fp32 summation order (the reference's or the HIP kernels') is from
 * the exact value.
 *
 * Layout (all contiguous fp32, NCHW), Hp = H+ks-1, Wp = W+ks-1:
 *   in  [B,C,Hp,Wp]   v,h [B,ks,H,W]   out,gO [B,C,H,W]
 *   gI  [B,C,Hp,Wp]   gV,gH [B,ks,H,W]
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Parity status: the reference holds no known-answer vectors for
 * this op (it has no tests at all); the restatement is pinned by analytic
 * known-answer tests (delta taps, box taps, adjoint identity, fp64 gradcheck)
 * in tests/test_oracle_sepconv.py.
 *
 * Threading: the outermost (batch x plane) loop is an OpenMP parallel-for so
 * that the timed CPU baseline can use the host cores; results do not depend on
 * the thread count (each output element is produced by exactly one thread with
 * the reference's serial summation order).
 */
#include <stddef.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define IN_AT(p, b, c, y, x) ((p)[(((size_t)(b) * C + (c)) * Hp + (y)) * Wp + (x)])
#define TAP_AT(p, b, f, y, x) ((p)[(((size_t)(b) * ks + (f)) * H + (y)) * W + (x)])
#define OUT_AT(p, b, c, y, x) ((p)[(((size_t)(b) * C + (c)) * H + (y)) * W + (x)])

int sepconv_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void sepconv_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* .cu:19-47 */
#define DEFINE_FORWARD(NAME, ACC)                                                            \
    void NAME(const float* in, const float* v, const float* h, float* out, int B, int C,     \
              int H, int W, int ks) {                                                        \
        const int Hp = H + ks - 1, Wp = W + ks - 1;                                          \
        _Pragma("omp parallel for collapse(2) schedule(static)")                             \
        for (int b = 0; b < B; ++b)                                                          \
            for (int c = 0; c < C; ++c)                                                      \
                for (int y = 0; y < H; ++y)                                                  \
                    for (int x = 0; x < W; ++x) {                                            \
                        ACC acc = 0;                                                         \
                        for (int fy = 0; fy < ks; ++fy)                                      \
                            for (int fx = 0; fx < ks; ++fx)                                  \
                                acc += (ACC)IN_AT(in, b, c, y + fy, x + fx) *                \
                                       (ACC)TAP_AT(v, b, fy, y, x) * (ACC)TAP_AT(h, b, fx, y, x); \
                        OUT_AT(out, b, c, y, x) = (float)acc;                                \
                    }                                                                        \
    }

/* .cu:49-86: thread index decodes to (b, fy, y, x) of grad_vertical */
#define DEFINE_GRAD_V(NAME, ACC)                                                             \
    void NAME(const float* gO, const float* in, const float* h, float* gV, int B, int C,     \
              int H, int W, int ks) {                                                        \
        const int Hp = H + ks - 1, Wp = W + ks - 1;                                          \
        _Pragma("omp parallel for collapse(2) schedule(static)")                             \
        for (int b = 0; b < B; ++b)                                                          \
            for (int fy = 0; fy < ks; ++fy)                                                  \
                for (int y = 0; y < H; ++y)                                                  \
                    for (int x = 0; x < W; ++x) {                                            \
                        ACC acc = 0;                                                         \
                        for (int c = 0; c < C; ++c)                                          \
                            for (int fx = 0; fx < ks; ++fx)                                  \
                                acc += (ACC)OUT_AT(gO, b, c, y, x) *                         \
                                       (ACC)IN_AT(in, b, c, y + fy, x + fx) *                \
                                       (ACC)TAP_AT(h, b, fx, y, x);                          \
                        TAP_AT(gV, b, fy, y, x) = (float)acc;                                \
                    }                                                                        \
    }

/* .cu:88-118: thread index decodes to (b, fx, y, x) of grad_horizontal */
#define DEFINE_GRAD_H(NAME, ACC)                                                             \
    void NAME(const float* gO, const float* in, const float* v, float* gH, int B, int C,     \
              int H, int W, int ks) {                                                        \
        const int Hp = H + ks - 1, Wp = W + ks - 1;                                          \
        _Pragma("omp parallel for collapse(2) schedule(static)")                             \
        for (int b = 0; b < B; ++b)                                                          \
            for (int fx = 0; fx < ks; ++fx)                                                  \
                for (int y = 0; y < H; ++y)                                                  \
                    for (int x = 0; x < W; ++x) {                                            \
                        ACC acc = 0;                                                         \
                        for (int c = 0; c < C; ++c)                                          \
                            for (int fy = 0; fy < ks; ++fy)                                  \
                                acc += (ACC)OUT_AT(gO, b, c, y, x) *                         \
                                       (ACC)IN_AT(in, b, c, y + fy, x + fx) *                \
                                       (ACC)TAP_AT(v, b, fy, y, x);                          \
                        TAP_AT(gH, b, fx, y, x) = (float)acc;                                \
                    }                                                                        \
    }

/* .cu:120-162: gather over the padded grid; loop order fx outer, fy inner;
 * (X, Y) = (xp - (ks-1) + jx, yp - (ks-1) + jy), taps indexed (ks-1) - j. */
#define DEFINE_GRAD_I(NAME, ACC)                                                             \
    void NAME(const float* gO, const float* v, const float* h, float* gI, int B, int C,      \
              int H, int W, int ks) {                                                        \
        const int Hp = H + ks - 1, Wp = W + ks - 1;                                          \
        _Pragma("omp parallel for collapse(2) schedule(static)")                             \
        for (int b = 0; b < B; ++b)                                                          \
            for (int c = 0; c < C; ++c)                                                      \
                for (int yp = 0; yp < Hp; ++yp)                                              \
                    for (int xp = 0; xp < Wp; ++xp) {                                        \
                        ACC acc = 0;                                                         \
                        for (int jx = 0; jx < ks; ++jx)                                      \
                            for (int jy = 0; jy < ks; ++jy) {                                \
                                const int X = xp - (ks - 1) + jx;                            \
                                const int Y = yp - (ks - 1) + jy;                            \
                                if (X < 0 || Y < 0 || Y >= H || X >= W) continue;            \
                                acc += (ACC)OUT_AT(gO, b, c, Y, X) *                         \
                                       (ACC)TAP_AT(v, b, (ks - 1) - jy, Y, X) *              \
                                       (ACC)TAP_AT(h, b, (ks - 1) - jx, Y, X);               \
                            }                                                                \
                        IN_AT(gI, b, c, yp, xp) = (float)acc;                                \
                    }                                                                        \
    }

DEFINE_FORWARD(sepconv_oracle_forward, float)
DEFINE_FORWARD(sepconv_oracle_forward_f64, double)
DEFINE_GRAD_V(sepconv_oracle_grad_v, float)
DEFINE_GRAD_V(sepconv_oracle_grad_v_f64, double)
DEFINE_GRAD_H(sepconv_oracle_grad_h, float)
DEFINE_GRAD_H(sepconv_oracle_grad_h_f64, double)
DEFINE_GRAD_I(sepconv_oracle_grad_i, float)
DEFINE_GRAD_I(sepconv_oracle_grad_i_f64, double)

/* .cu:187-242: V, then H, then I, like the reference's three launches. */
void sepconv_oracle_backward(const float* gO, const float* in, const float* v, const float* h,
                             float* gI, float* gV, float* gH, int B, int C, int H, int W, int ks) {
    sepconv_oracle_grad_v(gO, in, h, gV, B, C, H, W, ks);
    sepconv_oracle_grad_h(gO, in, v, gH, B, C, H, W, ks);
    sepconv_oracle_grad_i(gO, v, h, gI, B, C, H, W, ks);
}

void sepconv_oracle_backward_f64(const float* gO, const float* in, const float* v, const float* h,
                                 float* gI, float* gV, float* gH, int B, int C, int H, int W,
                                 int ks) {
    sepconv_oracle_grad_v_f64(gO, in, h, gV, B, C, H, W, ks);
    sepconv_oracle_grad_h_f64(gO, in, v, gH, B, C, H, W, ks);
    sepconv_oracle_grad_i_f64(gO, v, h, gI, B, C, H, W, ks);
}
