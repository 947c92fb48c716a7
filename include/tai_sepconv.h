/*
 * tai_sepconv.h -- C ABI of the MI355X-native adaptive separable convolution.
 *
 * Drop-in boundary for the reference's only native component.  Each entry point
 * replaces one symbol of the reference's cffi-exported C shim:
 *
 *   tai_sepconv_forward   <- int SeparableConvolution_cuda_forward(THCudaTensor* input, vertical,
 *                            horizontal, output, int ks)
 *                            src/separable_convolution/cfile/SeparableConvolution_cuda.h:1-7
 *                            (body cuda.c:8-25 -> launcher SeparableConvolution_kernel.cu:164-185)
 *   tai_sepconv_backward  <- int SeparableConvolution_cuda_backward(grad_output, input, vertical,
 *                            horizontal, grad_input, grad_vertical, grad_horizontal, int ks)
 *                            src/separable_convolution/cfile/SeparableConvolution_cuda.h:9-18
 *                            (body cuda.c:28-51 -> launcher SeparableConvolution_kernel.cu:187-242)
 *
 * THC tensor handles no longer exist (torch >= 1.0), so tensors cross the boundary as raw device
 * pointers plus their dimensions.  Conventions kept from the reference:
 *   - the CALLER owns and pre-allocates every buffer, outputs included; the callee only writes
 *     (SeparableConvolution.py:36,69-71).  Outputs need not be zeroed: every element is written.
 *   - all tensors are contiguous fp32 NCHW on the device the stream belongs to;
 *   - the call is asynchronous on `stream` (the reference used the THC current stream) and does
 *     no host synchronisation, allocation or copy, so it can be captured into a hipGraph.
 * Convention changed: the reference returned 1 unconditionally and reported launch failures through
 * THCudaCheck; these functions return TAI_SEPCONV_OK (0) or a negative TAI_SEPCONV_E* code, and
 * tai_sepconv_last_error() gives the text.
 *
 * Shapes (Hp = H + ks - 1, Wp = W + ks - 1; the reference asserts exactly this relation,
 * SeparableConvolution.py:27-29):
 *   input       [B, C, Hp, Wp]   replication-padded source frame
 *   vertical    [B, ks, H, W]    per-pixel vertical taps
 *   horizontal  [B, ks, H, W]    per-pixel horizontal taps
 *   output      [B, C, H, W]     out[b,c,y,x] = sum_fy sum_fx in[b,c,y+fy,x+fx] v[b,fy,y,x] h[b,fx,y,x]
 *   grad_*      same shapes as the tensor they are the gradient of.
 */
#ifndef TAI_SEPCONV_H
#define TAI_SEPCONV_H

#ifdef __cplusplus
extern "C" {
#endif

#define TAI_SEPCONV_OK 0
#define TAI_SEPCONV_EINVAL (-1)  /* null pointer, non-positive dimension, index space >= 2^31 */
#define TAI_SEPCONV_ELAUNCH (-2) /* hipGetLastError() after a launch was not hipSuccess */

/* Forward: replaces SeparableConvolution_cuda_forward (SeparableConvolution_cuda.h:1-7). */
int tai_sepconv_forward(const float* input, const float* vertical, const float* horizontal,
                        float* output, int B, int C, int H, int W, int ks, void* hip_stream);

/* Backward (all three gradients, in the reference's V, H, I order):
 * replaces SeparableConvolution_cuda_backward (SeparableConvolution_cuda.h:9-18).
 * Any of grad_input / grad_vertical / grad_horizontal may be NULL to skip that gradient. */
int tai_sepconv_backward(const float* grad_output, const float* input, const float* vertical,
                         const float* horizontal, float* grad_input, float* grad_vertical,
                         float* grad_horizontal, int B, int C, int H, int W, int ks,
                         void* hip_stream);

/* Bilinear x2 upsampling, align_corners = true, fp32 NCHW with planes = N*C: output [planes, 2H, 2W].
 * Replaces the THCUNN kernel behind the reference's torch.nn.Upsample(scale_factor=2, mode='bilinear') calls
 * (src/models/tai/tai.py:283,337,343; torch 0.3.1 semantics = align_corners=True), same caller-allocates / asynchronous-
 * on-stream conventions as above. */
int tai_upsample_bilinear2x_forward(const float* input, float* output, int planes, int H, int W, void* hip_stream);
/* Its gradient: grad_output [planes, 2H, 2W] -> grad_input [planes, H, W] (every element written), as a gather with the
 * forward's weights; sums in a fixed order (ATen's backward scatters with atomics). */
int tai_upsample_bilinear2x_backward(const float* grad_output, float* grad_input, int planes, int H, int W, void* hip_stream);

/* In-place x[n,c,:] = act(x[n,c,:] + bias[c]) over a contiguous fp32 [N, C, HW] tensor; act: 0 none, 1 ReLU, 2 tanh.
 * Finishes the bias-free MIOpen convolutions of the generator in one pass (the reference's nn.Conv2d + nn.ReLU / nn.Tanh
 * pairs, src/models/mcnet/mcnet.py:28-43,79-102,137-144,172-176,203-225; src/models/tai/tai.py:256-261). */
int tai_bias_act_inplace(float* x, const float* bias, int N, int C, int HW, int act, void* hip_stream);

/* out [planes, 2h, 2w] = res + fixed_unpooling(x), x [planes, h, w] landing on the even (2i, 2j) sites (DecCnn:
 * src/models/mcnet/mcnet.py:234-236, 240-256); fp32 contiguous, w even; out may not alias x or res. */
int tai_unpool2x_add(const float* x, const float* res, float* out, long long planes, int h, int w, void* hip_stream);

/* ConvLSTM gate arithmetic in one pass (ConvLstmCell.forward, src/models/mcnet/mcnet.py:287-293): gates [N, 4F, HW] holds the
 * chunks (i, j, f, o) of conv(cat(input, h)); c, new_c, new_h are [N, F, HW]; HW % 4 == 0.
 *   new_c = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j);   new_h = tanh(new_c) * sigmoid(o). */
int tai_convlstm_gates_forward(const float* gates, const float* c, float* new_c, float* new_h, int N, int F, int HW,
                               float forget_bias, void* hip_stream);
/* Its gradient: from dL/dnew_c and dL/dnew_h (either may be NULL: no gradient on that path) to grad_gates [N, 4F, HW] and
 * grad_c [N, F, HW]; gates, c and new_c as in the forward call. */
int tai_convlstm_gates_backward(const float* gates, const float* c, const float* new_c, const float* grad_new_c,
                                const float* grad_new_h, float* grad_gates, float* grad_c, int N, int F, int HW, float forget_bias,
                                void* hip_stream);

/* Direct "same"-padded stride-1 convolutions for the generator's thin layers, bias and activation fused (act: 0 none,
 * 1 ReLU, 2 tanh), fp32 NCHW contiguous, W % 4 == 0:
 *   tai_conv_cin1_forward       x [N,1,H,W], weight [Co,1,k,k] (k in {3,5}), y [N,Co,H,W]   (nn.Conv2d(1, gf, 5, padding=2)
 *                               + ReLU, src/models/mcnet/mcnet.py:28-31; nn.Conv2d(c_dim=1, gf, 3, padding=1) + ReLU, :79-81)
 *   tai_conv_cout1_3x3_forward  x [N,Ci,H,W], weight [1,Ci,3,3] in conv2d layout, y [N,1,H,W]
 *                               (nn.ConvTranspose2d(gf, c_dim=1, 3, padding=1) + Tanh, mcnet.py:223-224, after the
 *                               transpose-and-flip that turns it into a direct convolution). */
int tai_conv_cin1_forward(const float* x, const float* weight, const float* bias, float* y, int N, int Co, int H, int W,
                          int k, int act, void* hip_stream);
/* tai_conv_cin1_forward that also writes ypool [N,Co,H/2,W/2] = 2x2 max pool of the activated output (even H). */
int tai_conv_cin1_forward_maxpool(const float* x, const float* weight, const float* bias, float* y, float* ypool, int N, int Co,
                                  int H, int W, int k, int act, void* hip_stream);
/* ... with ypool written into a plane of pool_h x pool_w whose origin is at (pool_oy, pool_ox):
 * the halo-carrying input plane of the next layer (tai_conv3x3_wino_forward_ex, shift_k), halo left untouched. */
int tai_conv_cin1_forward_maxpool_window(const float* x, const float* weight, const float* bias, float* y, float* ypool, int N,
                                         int Co, int H, int W, int k, int act, int pool_h, int pool_w, int pool_oy, int pool_ox,
                                         void* hip_stream);
int tai_conv_cout1_3x3_forward(const float* x, const float* weight, const float* bias, float* y, int N, int Ci, int H,
                               int W, int act, void* hip_stream);
/* ... 5 x 5, padding 2, no activation, bias may be NULL: x [N,Ci,H,W], weight [1,Ci,5,5] -> y [N,1,H,W]; the input gradient of
 * tai_conv_cin1_forward (k = 5) when the input frame is itself generated (weight = the layer's filter flipped). */
int tai_conv_cout1_5x5_forward(const float* x, const float* weight, const float* bias, float* y, int N, int Ci, int H, int W,
                               void* hip_stream);

/* k x k (k = 5, 7) "same" convolution as a 3x3 convolution: out [N, S*S*C, H+2, W+4] (S = 2 for k = 5, 3 for k = 7)
 * receives the S*S shifted copies of x [N, C, H, W], each with its own halo -- out[n][(a*S+b)*C+c][u][v] =
 * x[n][c][u-1+3a-k/2+1][v-2+3b-k/2+1], zero outside the image -- and the k x k weight, cut into S x S blocks of 3 x 3 taps
 * ([K, S*S*C, 3, 3], zero past k), is then an ordinary weight for tai_conv3x3_wino_transform_weights; the convolution is
 * tai_conv3x3_wino_forward_window(out, ..., in_h = H+2, in_w = W+4, in_oy = 1, in_ox = 2).  Replaces
 * nn.Conv2d(gf, 2gf, 5, padding=2) and nn.Conv2d(2gf, 4gf, 7, padding=3) of MotionEnc (src/models/mcnet/mcnet.py:36-38,
 * 45-47).  W % 4 == 0. */
int tai_conv_shift_stack(const float* x, float* out, int N, int C, int H, int W, int k, void* hip_stream);

/* Element-wise tail of a discriminator layer evaluated on all sliding windows at once (nw window groups of B images stacked
 * along the batch; window t's weight is w0 * inv_scale[t], src/discriminators/SNDiscriminator.py:60-68, 140-159):
 *   forward, in place:  y = leaky_relu(y * inv_scale[window] + bias[channel], slope)          y [nw * B, C, HW], HW % 4 == 0
 *   backward:  grad_z = grad_y * (y > 0 ? 1 : slope)  (the pre-activation's gradient: weight / bias gradients take it),
 *              grad_scaled = grad_z * inv_scale[window]  (what flows into the input gradient through w0). */
int tai_window_scale_bias_lrelu(float* y, const float* bias, const float* inv_scale, int nw, int B, int C, int HW, float slope,
                                void* hip_stream);
int tai_window_scale_lrelu_backward(const float* grad_y, const float* y, const float* inv_scale, float* grad_z, float* grad_scaled,
                                    int nw, int B, int C, int HW, float slope, void* hip_stream);

/* Weight and bias gradients of the thin layers (tai_conv_cin1_forward / tai_conv_cout1_3x3_forward under loss.backward()):
 *   dw[cb][a][b] = sum over n, y, x of big[n, cb, y, x] * thin[n, 0, y + a - k/2, x + b - k/2]  (zero padding),  dbias[cb] = sum of big[n, cb]
 * big [N, Cb, H, W], thin [N, 1, H, W] fp32 contiguous, W % 4 == 0, k in {3, 5}; dw [Cb, k, k] or dbias [Cb] may be NULL;
 * workspace: N * Cb * 32 floats.  One-input-channel convolution: big = dL/dy, thin = x.  One-output-channel convolution:
 * big = x, thin = dL/dy, and dw[c][a][b] is the gradient of weight[0][c][k-1-a][k-1-b].  Reproducible (fixed summation order). */
int tai_thin_conv_wrw(const float* big, const float* thin, float* dw, float* dbias, float* workspace, int N, int Cb, int H, int W,
                      int k, void* hip_stream);

/* Activation + 2x2 max pool behind a convolution, training form (nn.ReLU + nn.MaxPool2d(2) of ContentEnc / MotionEnc,
 * src/models/mcnet/mcnet.py:28-60, 79-118): z, y [planes, H, W], ypool [planes, H/2, W/2] fp32 contiguous, H even, W % 4 == 0.
 *   forward:  y = relu ? max(z, 0) : z;  ypool = max over each 2x2 window of y                     (y may alias z)
 *   backward: grad_z = (relu ? [y > 0] : 1) * (grad_y + grad_ypool routed to the first maximum of its window in row-major
 *             order, nn.MaxPool2d's tie rule); grad_y or grad_ypool may be NULL (that path carries no gradient). */
int tai_act_maxpool2x2_forward(const float* z, float* y, float* ypool, long long planes, int H, int W, int relu, void* hip_stream);
int tai_act_maxpool2x2_backward(const float* grad_y, const float* grad_ypool, const float* y, float* grad_z, long long planes, int H,
                                int W, int relu, void* hip_stream);

/* Weight gradient of the 3x3 stride-1 padding-1 convolution y = conv(x, w) (the reference's nn.Conv2d / ConvTranspose2d
 * 3x3 layers under loss.backward(), src/environments/environments.py:348-355), in the Winograd domain on the fp32 MFMA pipe:
 *   dw [K, C, 3, 3] = sum over n, y, x of dy[n, k, y, x] * x[n, c, y + a - 1, x + b - 1]      (zero padding)
 * and, when dbias is not NULL, the bias gradient dbias [K] = sum over n, y, x of dy[n, k, y, x] (the output gradient
 * passes through the kernel anyway).
 * x [N, C, H, W], dy [N, K, H, W] fp32 contiguous, H even, W % 16 == 0, each tensor below 2 GiB.  workspace: device memory
 * of tai_conv3x3_wino_wrw_workspace_floats(...) floats (-1: shape not supported), overwritten.  Partial sums of the
 * workgroups are combined in a fixed order: the results are reproducible from call to call. */
long long tai_conv3x3_wino_wrw_workspace_floats(int N, int C, int K, int H, int W);
int tai_conv3x3_wino_wrw(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int N, int C, int K, int H,
                         int W, void* hip_stream);
/* ... with x given as a plane of in_h x in_w per channel whose pixel (in_oy, in_ox) lies under output pixel (0, 0): an input
 * that carries its own halo (the shifted-copy stack of the 5x5 / 7x7 layers, tai_conv_shift_stack: (H + 2, W + 4, 1, 2)) is
 * read inside it, zero padding applies outside the plane only.  dw is then the gradient of the blocked 3x3 weight. */
int tai_conv3x3_wino_wrw_window(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int N, int C, int K,
                                int H, int W, int in_h, int in_w, int in_oy, int in_ox, void* hip_stream);
/* Load scheme of the weight-gradient kernel when W % 32 == 0: 1 (default) = chunk pairs over 16 consecutive tiles, whole
 * 128-byte lines per load; 0 = 8-tile chunks as for the other widths (same results, for A/B timing).  Returns the previous value. */
int tai_conv3x3_wino_wrw_set_paired(int on);
/* Transform domain of tai_conv3x3_wino_wrw, process-wide: 2 = F(2x2, 3x3) (csrc/wino_wrw.hip.inc), 4 (default) = F(4x4, 3x3) where H % 4 == 0 (and
 * W % 16 == 0, no input window) -- 36 multiplies per 4 x 4 tile instead of 16 per 2 x 2 tile, the forward kernel's interpolation points
 * (csrc/wino43_conv.hip.inc, conv3x3_wrw_gen); other shapes keep the F(2x2, 3x3) kernel.  Same interface, same workspace (the size query
 * covers both), reproducible either way.  Returns the previous value, -1 for anything else. */
int tai_conv3x3_wino_wrw_set_tile(int tile);

/* Spectral normalisation of one discriminator layer, as the reference's SNConv2d / SNLinear do on every forward
 * (src/discriminators/SNDiscriminator.py:10-25 max_singular_value, :60-68 and :84-92 W.data <- W.data / sigma):
 * Ip rounds of  v <- normalise(u W), u <- normalise(v W^T)  on weight [out_rows, in_cols] (the layer's weight viewed as a
 * matrix, fp32 contiguous), sigma = (v W^T) u^T, then weight /= sigma IN PLACE and u [out_rows] replaced by the new
 * vector.  scratch: at least in_cols + out_rows floats of device memory.  sigma_out: one float of device memory that
 * receives sigma (may be NULL) -- a caller that renormalises the same layer several times in a row (the discriminator's
 * sliding windows) passes consecutive slots of one vector and gets the cumulative scale of every window from it.
 * 2 Ip + 1 kernel launches on hip_stream, nothing synchronises. */
int tai_sn_power_iteration(float* weight, float* u, float* scratch, float* sigma_out, int out_rows, int in_cols, int Ip,
                           void* hip_stream);

/* 3x3 stride-1 zero-padded ("same") convolution + bias + activation, fp32 NCHW contiguous, H and W even, computed as
 * Winograd F(2x2,3x3) on the fp32 MFMA pipe.  Replaces nn.Conv2d(C, K, 3, padding=1) [+ ReLU] of the generator and the
 * kernel network (src/models/mcnet/mcnet.py:79-118,131-152,165-170,271; src/models/tai/tai.py:248-286) and, after the
 * transpose-and-flip of the weight, nn.ConvTranspose2d(C, K, 3, padding=1) of DecCnn (mcnet.py:198-224).
 *   tai_conv3x3_wino_weight_floats      number of floats of the transformed-weight buffer U for a [K,C,3,3] weight
 *   tai_conv3x3_wino_transform_weights  weight [K,C,3,3] -> U (once per weight; U is what the forward reads)
 *   tai_conv3x3_wino_forward            x [N,C,H,W], U, bias [K] -> y [N,K,H,W]; act: 0 none, 1 ReLU, 2 tanh */
long long tai_conv3x3_wino_weight_floats(int K, int C);
int tai_conv3x3_wino_transform_weights(const float* weight, float* U, int K, int C, void* hip_stream);
int tai_conv3x3_wino_forward(const float* x, const float* U, const float* bias, float* y, int N, int C, int K, int H, int W,
                             int act, void* hip_stream);
/* The same convolution, also writing ypool [N,K,H/2,W/2] = 2x2 max pool of the activated output (nn.Conv2d + ReLU +
 * nn.MaxPool2d(2): src/models/mcnet/mcnet.py:84-88, 96-100, 110-114; the un-pooled output is the residual, :118). */
int tai_conv3x3_wino_forward_maxpool(const float* x, const float* U, const float* bias, float* y, float* ypool, int N, int C,
                                     int K, int H, int W, int act, void* hip_stream);
/* The same convolution on an input plane of in_h x in_w that holds output pixel (0, 0) at (in_oy, in_ox) (in_ox and in_w
 * even): the zero padding applies outside that plane only, so an input that carries its own halo (tai_conv_shift_stack)
 * is convolved without padding.  ypool may be NULL (no pooled output). */
int tai_conv3x3_wino_forward_window(const float* x, const float* U, const float* bias, float* y, float* ypool, int N, int C,
                                    int K, int H, int W, int in_h, int in_w, int in_oy, int in_ox, int act, void* hip_stream);
/* The same convolution with the input given as `nparts` (1..4) contiguous [N, C / nparts, H, W] tensors, the operands of
 * a torch.cat along the channels that is then never materialised (Residual: src/models/mcnet/mcnet.py:182, CombLayers
 * :152, TAI.forward: src/models/tai/tai.py:188).  xs: host array of device pointers; C / nparts must be a multiple of 8. */
int tai_conv3x3_wino_forward_parts(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N,
                                   int C, int K, int H, int W, int act, void* hip_stream);
/* The general form of the same convolution (every optional argument may be NULL / 0):
 *   xs, nparts      1..4 input parts as above; or ONE tensor with shift_k = k, the size of a k x k filter (4 <= k <= 9):
 *                   with S = (k + 2) / 3 the input [N, C / S^2, in_h, in_w] is read S x S times, channel block (a, b) =
 *                   block a * S + b displaced by (3a, 3b) pixels, and the weight is the k x k filter cut into S x S blocks
 *                   of 3 x 3 taps, ZERO PAST k ([K, S^2 * Cin, 3, 3]).  This is the k x k "same" convolution of MotionEnc
 *                   (nn.Conv2d(gf, 2gf, 5, padding=2), nn.Conv2d(2gf, 4gf, 7, padding=3): src/models/mcnet/mcnet.py:36-38,
 *                   45-47) read from the pooled output of the layer before it where it lies -- no stack of shifted copies.
 *                   When k is not a multiple of 3 the last block row / column has an all-zero third tap row / column, hence
 *                   an all-zero fourth row / column of transformed weights: those multiply-adds are skipped (16 % of a
 *                   7 x 7 layer's, 23 % of a 5 x 5 layer's).  The plane must carry the halo: image row 0 at row k/2, so
 *                   in_oy = 1; image column 0 at column in_ox + k/2 - 1 with in_ox even and >= 2;
 *                   in_h >= H + in_oy + 1 + 3 (S - 1), in_w >= W + in_ox + 2 + 3 (S - 1), halo zero.
 *   ypool, pool_*   2x2 max pool of the activated output written into a plane of pool_h x pool_w with its origin at
 *                   (pool_oy, pool_ox) (pool_h = 0: a plain [N, K, H/2, W/2] tensor);
 *   addx, y2        y2 [N, K, H, W] = y + fixed_unpooling(addx), addx [N, K, H/2, W/2] landing on the even (2i, 2j) sites:
 *                   DecCnn's unpool + residual add (src/models/mcnet/mcnet.py:234-236, 240-256) as a second output of the
 *                   Residual block's last convolution (mcnet.py:172-176).  With addx and y2 == NULL the sum is written to y
 *                   (the plain convolution output is then not produced). */
int tai_conv3x3_wino_forward_ex(const float* const* xs, int nparts, int shift_k, const float* U, const float* bias, float* y,
                                float* ypool, int pool_h, int pool_w, int pool_oy, int pool_ox, const float* addx, float* y2, int N,
                                int C, int K, int H, int W, int in_h, int in_w, int in_oy, int in_ox, int act, void* hip_stream);
/* The same convolution as Winograd F(4x4, 3x3) on the fp32 MFMA pipe (csrc/wino43_conv.hip.inc): 36 multiplies per 4 x 4 output tile
 * instead of 16 per 2 x 2 -- 1.78x fewer MFMAs, ~7x the fp32 rounding error per layer; meant for the layers with C >= 128 and
 * K >= 128, where the bi-TAI forward's end-to-end error is unchanged (profiles/r04_wino_f43_study.txt, r04_wino43_default_parity.txt):
 * the Python side (conv_ops) sends exactly those layers here by default.  Own transformed-weight
 * layout (tai_conv3x3_wino43_weight_floats / _transform_weights); H and W multiples of 4; C any (the transformed weights are zero-padded
 * to a multiple of 4 channels); act as above.  Replaces the same
 * reference layers as tai_conv3x3_wino_forward (src/models/mcnet/mcnet.py:79-118,131-152,165-176,198-224). */
long long tai_conv3x3_wino43_weight_floats(int K, int C);
int tai_conv3x3_wino43_transform_weights(const float* weight, float* U, int K, int C, void* hip_stream);
int tai_conv3x3_wino43_forward(const float* x, const float* U, const float* bias, float* y, int N, int C, int K, int H, int W, int act,
                               void* hip_stream);
/* ... the general form: 1 to 4 input parts and the second outputs of tai_conv3x3_wino_forward_ex that a 4 x 4 tile can hold --
 *   ypool [N,K,H/2,W/2]  the 2x2 max pool of the activated output (act 0 or 1), or NULL;
 *   addx  [N,K,H/2,W/2]  y2 = y + fixed_unpooling(addx) (act 0, no ypool); with y2 NULL the sum is written to y and the plain output
 *                        is not produced. */
int tai_conv3x3_wino43_forward_ex(const float* const* xs, int nparts, const float* U, const float* bias, float* y, float* ypool,
                                  const float* addx, float* y2, int N, int C, int K, int H, int W, int act, void* hip_stream);
/* The k x k "same" convolutions of MotionEnc (nn.Conv2d(gf, 2gf, 5, padding=2), nn.Conv2d(2gf, 4gf, 7, padding=3):
 * src/models/mcnet/mcnet.py:36-38, 45-47) on the same kernel: tai_conv3x3_wino_forward_ex's displaced-read form (shift_k = k, S = (k + 2) / 3,
 * x ONE plane [N, C / S^2, in_h, in_w] that carries its halo and is read S x S times, channel block (a, b) displaced by (3a, 3b) pixels;
 * U from the k x k filter cut into S x S blocks of 3 x 3 taps, zero past k: [K, S^2 * Cin, 3, 3] through
 * tai_conv3x3_wino43_transform_weights) with the 4 x 4 tile: 1.78x fewer MFMAs than there.  The pixel under output (0, 0)'s centre tap of
 * block (0, 0) is at (in_oy, in_ox) (any in_ox >= 1: the patch rows are loaded 4-byte aligned); the plane must cover rows
 * in_oy - 1 ... in_oy + H + 3 (S - 1) and columns in_ox - 1 ... in_ox + W + 3 (S - 1), halo zero.  ypool (may be NULL): 2x2 max pool of
 * the activated output, into a plane of pool_h x pool_w with its origin at (pool_oy, pool_ox) (pool_w, pool_ox even; pool_h = 0: a plain
 * [N, K, H/2, W/2] tensor).  act 0 / 1; C / S^2, H, W multiples of 4. */
int tai_conv3x3_wino43_forward_blocks(const float* x, int shift_k, const float* U, const float* bias, float* y, float* ypool, int pool_h,
                                      int pool_w, int pool_oy, int pool_ox, int N, int C, int K, int H, int W, int in_h, int in_w, int in_oy,
                                      int in_ox, int act, void* hip_stream);
/* Kept for the tools build: 0 = the kernel; values 101-112 select timing ablations / schedule variants of its generated chunk loop where the
 * library was built with -DTAI_TIMING_VARIANTS (wrong results by design).  Returns the previous value, -1 on a value this build does not have
 * (round 4's compiler-scheduled forms 8 / 4 are no longer in the library). */
int tai_conv3x3_wino43_set_waves(int waves);
/* Workgroup placement of the F(4x4, 3x3) kernels (forward, blocks, weight gradient), process-wide: 1 (default) = aware of the chip's 8 XCDs
 * (the hardware deals consecutive workgroups to them in turn, each with its own L2): an XCD gets one output-channel block and a contiguous
 * run of tile blocks (forward) / whole splits (weight gradient); 0 = the plain dispatch order of rounds 4-5.  Same results; for A/B timing.
 * Returns the previous value. */
int tai_conv3x3_wino43_set_placement(int xcd_aware);
/* ... with the input given as 1 to 4 equal channel parts (contiguous [N, C / nparts, H, W] tensors; C / nparts a multiple of 4): the
 * operands of a torch.cat along the channels that is never materialised (tai_conv3x3_wino_forward_parts' counterpart). */
int tai_conv3x3_wino43_forward_parts(const float* const* xs, int nparts, const float* U, const float* bias, float* y, int N, int C, int K,
                                     int H, int W, int act, void* hip_stream);
/* Arithmetic of the Winograd GEMMs, process-wide.  0 (default): fp32 MFMA -- the reference's arithmetic class (cuDNN fp32 behind
 * nn.Conv2d, src/models/mcnet/mcnet.py:28-224) and the one every parity statement of this library is made on.  1 (opt-in): SPLIT
 * bf16 -- each fp32 operand as three bf16 terms, a product as six bf16 products accumulated in fp32 on the bf16 MFMA pipe
 * (csrc/wino_split.hip.inc); its error against a float64 network is at or below the fp32 form's on every bi-TAI layer
 * (profiles/r04_split_bf16_study.txt).  The mode decides what tai_conv3x3_wino_weight_floats / _transform_weights produce (mode 1:
 * the fp32 image followed by the split image); the forward entry points follow the buffer they are handed, so a buffer is always
 * read in the layout it was written in, and layers the split kernel does not take (displaced reads, tile rows that are neither a
 * power of two nor a multiple of 16 tiles) run the fp32 kernel from the same buffer.  Returns the previous mode, negative on a bad
 * argument. */
int tai_conv3x3_wino_set_arithmetic(int mode);
int tai_conv3x3_wino_get_arithmetic(void);
/* The library records which buffers tai_conv3x3_wino_transform_weights filled with a split-bf16 image (the forward entry points follow the
 * buffer they are handed).  Call this when such a buffer is freed, so that its address can never be read in a layout it no longer has
 * (returns 1 if a record was dropped, 0 if there was none). */
int tai_conv3x3_wino_forget_weights(const float* U);
/* Benchmarking: 0 keeps every layer on the 64-channel x 64-tile workgroup shape; 1 (default) lets layers whose K is a
 * multiple of 128 use the 128 x 32 shape.  Returns the previous value. */
int tai_conv3x3_wino_set_tall(int on);
/* Diagnostics (tools/wino_timeline.py): the same launch with ReLU; every workgroup also writes shader-clock stamps to
 * stamps[64 * workgroup + i]: i = 0 entry, 1 prologue done, 2 channel loop done, 3 end, 4 + c end of chunk c (c < 26),
 * and after tai_conv3x3_wino_timeline_skip(7) 30 + 16 * c + g end of MFMA group g of chunk c (c < 2).  stamps holds
 * 64 * workgroups int64.  tai_conv3x3_wino_timeline_skip(level): 0 the full kernel; 1, 2, 5 leave parts of it out to
 * time what remains (results are then wrong); affects timeline launches only, and only in the tools build of the library
 * (-DTAI_TIMING_VARIANTS): the shipped library accepts level 0 alone. */
int tai_conv3x3_wino_forward_timeline(const float* x, const float* U, const float* bias, float* y, int N, int C, int K, int H,
                                      int W, long long* stamps, void* hip_stream);
int tai_conv3x3_wino_timeline_skip(int level);

/* Selects a kernel variant for tai_sepconv_forward (benchmarking / tests):
 *   0 = automatic (default), 1 = generic one-thread-per-output kernel (any shape),
 *   2 = LDS-tiled, whole tap set register-resident, 3 = LDS-tiled, taps split over half-waves,
 *   4 = LDS-tiled, register-resident taps, packed fp32 FMAs (v_pk_fma_f32),
 *   5 = as 4 with the row loop hand-scheduled in gfx950 assembly (v planes by LDS-DMA), 6 = as 5 with
 *       the tap loads of half the waves deferred behind a workgroup barrier, 7-9 = 16-row tiles (8 waves),
 *   10-13 = 8-wave workgroups mixing "taps first" (type A) and "taps last" (type B) waves on every SIMD;
 *       13 adds 16-byte patch staging and alternating wave priorities, 16 = 13 with each XCD given a contiguous
 *       eighth of the tile list, 18 = 16 with the type-A tap loads issued at kernel entry: the patch is staged by the
 *       type-B waves alone (LDS-DMA) and announced through an LDS counter instead of a workgroup barrier,
 *   20 (default for C == 1) = kernel 18's wave types as ONE persistent workgroup per CU when the launch has more tiles than
 *       the device has CUs (and a multiple of 8 of them): the next tile's patch and taps are on their way while this tile
 *       computes; with at most one tile per CU it IS kernel 18 (selected explicitly, 20 runs persistent at any tile count),
 *   14/15 = type-A waves that load their taps once and run the row loop once per channel (8- / 4-wave workgroups),
 *   17 = type-A waves that walk three channel patches per tap row: v and h are read once for all three channels
 *       (channels beyond a multiple of three run on the single-channel kernel), 19 (default for C > 1) = 17 with the three
 *       patches staged by LDS-DMA and the tap loads issued right behind them (patch and h stream travel together).
 *   Values >= 100 (timing experiments that produce wrong results) exist only in the tools build of the library
 *   (-DTAI_TIMING_VARIANTS, build/libtai_sepconv_timing.so); the shipped library rejects them with TAI_SEPCONV_EINVAL.
 * Returns the previous value. */
int tai_sepconv_set_forward_variant(int variant);
/* The variant `0 = automatic` resolves to for a frame of C channels, width W and filter size ks. */
int tai_sepconv_default_forward_variant(int C, int W, int ks);

/* grad_input kernel of tai_sepconv_backward:
 *   0 = automatic: wave-private accumulation strips + fixed-order slab sum when ks == 51, W % 4 == 0, C in {1, 3} --
 *       bit-reproducible; the tile slabs borrow the caller's grad_vertical (or grad_horizontal) buffer before that gradient
 *       is written, so gI is computed FIRST (with neither buffer given, the strips flush with float atomics instead);
 *       any other shape: the bounds-checked gather;
 *   1 = bounds-checked gather of the reference (any shape, bit-reproducible, ~40x slower);
 *   2 = round-1 kernel: LDS row-scatter with a barrier per tap row and float atomics (last bits depend on arrival order);
 *   3 = as 0 (explicit);  4 = as 0 with the row loop in HIP C++ instead of the generated assembly (A/B).
 * Returns the previous value. */
int tai_sepconv_set_grad_input_variant(int variant);

/* grad_vertical / grad_horizontal kernels: 0 = automatic (one fused launch of the hand-scheduled wave types when C == 1
 * and both are requested; the gV waves' tap loads issued at kernel entry, the patch staged by the gH waves through LDS-DMA),
 * 1 = the two separate HIP kernels, 2 = the fused launch with the patch staged behind a workgroup barrier first (round 2's
 * form, A/B), 3 / 4 = as 0 with the gV waves at priority 0 / 2 instead of their gH partners' 1 (A/B; the results are the same
 * bits).  Returns the previous value. */
int tai_sepconv_set_grad_taps_variant(int variant);

/* Algorithmic HBM bytes of one call: each operand read once, each result written once. */
long long tai_sepconv_forward_bytes(int B, int C, int H, int W, int ks);
long long tai_sepconv_backward_bytes(int B, int C, int H, int W, int ks);

/* Measurement utility (bench.py): one in-order streaming read of `bytes` (>= 1 MiB, 16-byte aligned) of device memory, 16 bytes
 * per lane with eight loads in flight, default cache policy (nt == 0) or non-temporal loads (nt != 0); `sink` holds 4096 floats and
 * is not written for ordinary data.  Asynchronous on the stream; time it with events.  No counterpart in the reference: it is the
 * yardstick the separable convolution's in-model launch (SeparableConvolution_kernel.cu:19-47 over 1.07 GB of once-read taps) is
 * held against on the box it runs on. */
int tai_hbm_read_probe(const void* buffer, long long bytes, int nt, float* sink, void* hip_stream);

/* Text of the last error on the calling thread ("" if none). */
const char* tai_sepconv_last_error(void);

/* Library / ABI version: major*10000 + minor*100 + patch. */
int tai_sepconv_version(void);

/* SHA-256 (hex) of the sources this binary was compiled from: every csrc/*.hip and csrc/*.inc plus this header, in sorted
 * order, as the in-tree builder computes it (video-frame-inpainting_amd/_native.py: source_hash()).  The loader recomputes
 * the hash from the tree next to the binary and refuses a library that was built from other sources -- the sources travel
 * with the binary, so a stale .so (a failed rebuild, an edited kernel) cannot run.  Replaces the unconditional load of
 * src/separable_convolution/_ext/cunnex/__init__.py:6-15.  "unknown" for a build made without the in-tree builder. */
const char* tai_sepconv_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* TAI_SEPCONV_H */
