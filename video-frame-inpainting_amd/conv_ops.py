"""Convolution + bias + activation as the generator uses it (reference: nn.Conv2d / nn.ConvTranspose2d followed by
nn.ReLU, nn.Tanh or nothing).  Inference on the GPU runs the bias-free MIOpen convolution and finishes it with the
single-pass HIP bias+activation kernel of the C ABI (ATen would add the bias and apply the ReLU in two more full passes);
anything that needs autograd, and CPU tensors, take the stock PyTorch ops -- same arithmetic, one rounding per add."""
import ctypes

import torch
import torch.nn.functional as F

from . import _native

_ACT = {None: 0, 'relu': 1, 'tanh': 2}


def _finish(y, bias, act):
    if act == 'relu':
        return torch.relu(y + bias.view(1, -1, 1, 1)) if bias is not None else torch.relu(y)
    if act == 'tanh':
        return torch.tanh(y + bias.view(1, -1, 1, 1)) if bias is not None else torch.tanh(y)
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


def _as_conv_weight(weight, transposed):
    # ConvTranspose2d(cin, cout, 3, stride 1, padding 1)  ==  conv2d with weight[o, i, ky, kx] = wt[i, o, 2-ky, 2-kx]
    return weight.transpose(0, 1).flip(2, 3) if transposed else weight


_EPOCH = [0]


def invalidate_derived(module=None):
    """Drop the cached derived weights (Winograd-domain U, direct-conv layout of a transposed conv) of ``module``'s
    parameters, or of every weight when ``module`` is None.  The cache notices writes that move the tensor's version
    counter (optimizer steps, ``load_state_dict``, ``with torch.no_grad(): p.mul_(2)``) by itself; a write through
    ``p.data`` moves neither the counter nor the pointer (``p.data.normal_()``, ``dist.broadcast(p.data, ...)``, a user's
    EMA or clipping), so whoever writes that way calls this afterwards -- ``util.weights_init``,
    ``parallel.broadcast_module_state`` and the environments' ``load`` do."""
    if module is None:
        _EPOCH[0] += 1
        return
    for p in module.parameters():
        if hasattr(p, '_tai_derived'):
            p._tai_derived.clear()


WINOGRAD_ARITHMETICS = {'fp32': 0, 'bf16x3': 1}
_WINO_ARITH = [0]          # mirrors the library's mode: part of the cache tag of every F(2x2, 3x3) transformed-weight buffer


def set_winograd_arithmetic(name):
    """Arithmetic of the Winograd 3x3 GEMMs (``tai_conv3x3_wino_set_arithmetic``): ``'fp32'`` (default: the fp32 MFMA, the
    reference's arithmetic class and the one every parity statement is made on) or ``'bf16x3'`` (opt-in: every fp32 operand
    as three bf16 terms, six bf16 products per product, fp32 accumulation -- csrc/wino_split.hip.inc).  The transformed
    weights of the two arithmetics are cached side by side (the arithmetic is part of the cache tag), so a hipGraph captured
    before a switch keeps replaying the arithmetic it was captured with ON BUFFERS THAT STAY ALIVE (the switch used
    to drop the other arithmetic's buffers, which a captured graph would then have read after free).  Returns the previous name."""
    mode = WINOGRAD_ARITHMETICS[name]
    prev = _native.lib().tai_conv3x3_wino_set_arithmetic(mode)
    if prev < 0:
        raise ValueError(name)
    _WINO_ARITH[0] = mode
    return [k for k, v in WINOGRAD_ARITHMETICS.items() if v == prev][0]


def get_winograd_arithmetic():
    mode = _native.lib().tai_conv3x3_wino_get_arithmetic()
    return [k for k, v in WINOGRAD_ARITHMETICS.items() if v == mode][0]


_WINO_TILE = [4]
WINO43_MIN_CHANNELS = 64         # F(4x4, 3x3) where both C and K are at least this: every 3x3 layer of MC-Net but the first and the last.  (128 until
                                 # the kernel moved from Lavin's interpolation points to (0, +-3/4, +-3/2, inf), whose per-layer rounding is ~4x lower:
                                 # with it the end-to-end error against the CPU oracle is the F(2x2) one on every config at 64 as well --
                                 # profiles/r05_wino_f43_points_study.txt, r05_wino43_min_channels.txt; 49.7 -> 46.75 ms on the configs[1] forward)
WINO43_MIN_WORKGROUPS = 150      # ... and where its 64-channel x 32-tile workgroups occupy most of the chip (round 5's kernel, same box, the
                                 # replayed configs[1] forward: 400: 51.11 ms, 256: 50.78, 150: 50.57, 100: 51.32, 64: 51.75 -- tools/w43_threshold_ab.py)


def set_weight_gradient_tile(m):
    """Transform domain of the 3x3 weight-gradient kernel (wino_weight_grad): 4 (default): F(4x4, 3x3) -- csrc/wino43_conv.hip.inc,
    conv3x3_wrw_gen: the forward kernel's chunk loop with the tiles as the reduction, 1.78x fewer MFMAs, 1.15-1.5x faster per layer -- where
    H % 4 == 0 and the input is a plain tensor; 2: F(2x2, 3x3) everywhere (csrc/wino_wrw.hip.inc, rounds 2-5).  Both sum in a fixed order.
    Returns the previous value."""
    prev = _native.lib().tai_conv3x3_wino_wrw_set_tile(m)
    if prev < 0:
        raise ValueError(m)
    return prev


def set_winograd_tile(m):
    """Output tile of the Winograd 3x3 convolutions: 4 (default since the end of round 4): F(4x4, 3x3), csrc/wino43_conv.hip.inc, on the
    layers with C >= 64 and K >= 64 (WINO43_MIN_CHANNELS; the layers marked by mark_outside_recurrence at any width) and enough workgroups
    -- 1.78x fewer MFMAs, fp32 operands on the fp32 MFMA as everywhere else; on its interpolation points (0, +-3/4, +-3/2, inf) the
    rounding error is ~3-4x F(2x2, 3x3)'s per layer and the forward's end-to-end error against the CPU oracle is that of F(2x2, 3x3) on
    every config (profiles/r05_wino_f43_points_study.txt, r05_wino43_min_channels.txt) -- and F(2x2, 3x3), csrc/wino_conv.hip.inc, on
    every other layer; or 2: F(2x2, 3x3) on every layer (the arithmetic of rounds 1-3).  A hipGraph captured before a switch
    keeps replaying what it captured.  Returns the previous value."""
    if m not in (2, 4):
        raise ValueError(m)
    prev, _WINO_TILE[0] = _WINO_TILE[0], m
    return prev


def get_winograd_tile():
    return _WINO_TILE[0]


def mark_outside_recurrence(module):
    """Tell the dispatch that ``module``'s 3x3 layers may take F(4x4, 3x3) below WINO43_MIN_CHANNELS too (from 16 input channels).  The
    kernel network and the merge residuals feed the separable convolution once per output frame, outside MC-Net's recurrence (where
    every prediction is the next step's input): with all of them on the 4 x 4 tile the forward's end-to-end error against float64 is
    unchanged (profiles/r05_wino_f43_policy_study.txt, measured with Lavin's points, whose rounding was ~4x the present ones').
    TAIFillInModel marks kernelnet and merge_residual{1,2,3} (tai.py)."""
    for p in module.parameters():
        p._tai_f43_any_width = True


def _wino43_ok(N, Ci, Co, H, W, nparts=1, weight=None):
    wide = (Ci >= WINO43_MIN_CHANNELS and Co >= WINO43_MIN_CHANNELS) or (Ci >= 16 and getattr(weight, '_tai_f43_any_width', False))
    return (_WINO_TILE[0] == 4 and wide and H % 4 == 0 and W % 4 == 0
            and Ci % nparts == 0 and ((Ci // nparts) % 4 == 0 or nparts == 1) and N * max(Ci, Co) * H * W < 2 ** 29
            and ((N * (H // 4) * (W // 4) + 31) // 32) * ((Co + 63) // 64) >= WINO43_MIN_WORKGROUPS)


def _cached(weight, tag, make):
    """Derived form of a weight, kept on the tensor object and rebuilt when the weight's version counter or storage
    moves, or after ``invalidate_derived``."""
    cache = getattr(weight, '_tai_derived', None)
    if cache is None:
        cache = {}
        weight._tai_derived = cache
    hit = cache.get(tag)
    key = (weight._version, weight.data_ptr(), _EPOCH[0])
    if hit is None or hit[0] != key:
        hit = (key, make())
        cache[tag] = hit
    return hit[1]


def _registered(U):
    """A transformed-weight buffer of the F(2x2, 3x3) entry points: the library keeps the addresses of the split-bf16 images it wrote (the
    forward entry points follow the buffer they are handed); the entry goes when the tensor does (the registry used to grow
    without bound and a recycled address could be read in the wrong layout)."""
    import weakref
    weakref.finalize(U, _native.lib().tai_conv3x3_wino_forget_weights, U.data_ptr())
    return U


def _wino_weights(weight, transposed):
    def make():
        w = _as_conv_weight(weight.detach(), transposed).contiguous()
        K, C = w.shape[0], w.shape[1]
        L = _native.lib()
        U = _registered(torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), dtype=torch.float32, device=w.device))
        with torch.cuda.device(w.device):
            _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C,
                                                               torch.cuda.current_stream(w.device).cuda_stream),
                          'tai_conv3x3_wino_transform_weights')
        return U
    return _cached(weight, ('wino', transposed, _WINO_ARITH[0]), make)


def _wino43_weights(weight, transposed):
    def make():
        w = _as_conv_weight(weight.detach(), transposed).contiguous()
        K, C = w.shape[0], w.shape[1]
        L = _native.lib()
        U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, C), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _native.check(L.tai_conv3x3_wino43_transform_weights(w.data_ptr(), U.data_ptr(), K, C,
                                                                 torch.cuda.current_stream(w.device).cuda_stream),
                          'tai_conv3x3_wino43_transform_weights')
        return U
    return _cached(weight, ('wino43', transposed), make)


def _block3x3_weight(weight):
    """[K, C, k, k] (k = 5, 7) -> [K, S*S*C, 3, 3]: block (a, b) of 3x3 taps of the k x k filter, zero past k, for the
    stack of shifted inputs tai_conv_shift_stack builds (channel order (a*S + b)*C + c)."""
    K, C, k, _ = weight.shape
    S = (k + 2) // 3
    wp = F.pad(weight, (0, 3 * S - k, 0, 3 * S - k))                        # [K, C, 3S, 3S]
    wp = wp.view(K, C, S, 3, S, 3).permute(0, 2, 4, 1, 3, 5)                # [K, a, b, C, 3, 3]
    return wp.reshape(K, S * S * C, 3, 3).contiguous()


def _wino_weights_kxk(weight, transposed=False):
    """``transposed``: the filter of the input-gradient convolution, weight[k, c, a, b] -> [c, k, K-1-a, K-1-b]."""
    def make():
        w = weight.detach()
        w = _block3x3_weight(w.transpose(0, 1).flip(2, 3).contiguous() if transposed else w)
        K, C = w.shape[0], w.shape[1]
        L = _native.lib()
        U = _registered(torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), dtype=torch.float32, device=w.device))
        with torch.cuda.device(w.device):
            _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C,
                                                               torch.cuda.current_stream(w.device).cuda_stream),
                          'tai_conv3x3_wino_transform_weights')
        return U
    return _cached(weight, ('wino_kxk', transposed, _WINO_ARITH[0]), make)


def _wino43_weights_kxk(weight, transposed=False):
    """The k x k filter as S x S blocks of 3 x 3 taps in the F(4x4, 3x3) layout (tai_conv3x3_wino43_forward_blocks); ``transposed``: the
    filter of the input-gradient convolution, weight[k, c, a, b] -> [c, k, K-1-a, K-1-b]."""
    def make():
        w = weight.detach()
        w = _block3x3_weight(w.transpose(0, 1).flip(2, 3).contiguous() if transposed else w)
        K, C = w.shape[0], w.shape[1]
        L = _native.lib()
        U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, C), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _native.check(L.tai_conv3x3_wino43_transform_weights(w.data_ptr(), U.data_ptr(), K, C,
                                                                 torch.cuda.current_stream(w.device).cuda_stream),
                          'tai_conv3x3_wino43_transform_weights')
        return U
    return _cached(weight, ('wino43_kxk', transposed), make)


def _wino43_blocks_ok(N, Cin, Co, H, W):
    """MotionEnc's 5x5 / 7x7 layers on the 4 x 4 tile (displaced reads): the 7x7 (128 -> 256) is a wide layer, the 5x5 (64 -> 128) was
    measured parity-neutral too (profiles/r05_wino_f43_policy_study.txt: "big + MotionEnc 5x5")."""
    return (_WINO_TILE[0] == 4 and Cin % 4 == 0 and Cin >= 16 and H % 4 == 0 and W % 4 == 0
            and ((N * (H // 4) * (W // 4) + 31) // 32) * ((Co + 63) // 64) >= WINO43_MIN_WORKGROUPS)


def _kxk_as_blocks(x, weight, bias, act, transposed=False):
    """5x5 / 7x7 "same" convolution on the 4 x 4 tile: x copied into a zero-haloed plane, read S x S times by displaced 3 x 3 blocks
    (tai_conv3x3_wino43_forward_blocks).  The training form's forward and input gradient (``transposed``)."""
    N, C, H, W = x.shape
    K, k = weight.shape[1 if transposed else 0], weight.shape[2]
    S, top, left, in_h, in_w = halo_geometry(H, W, k)
    plane = halo_plane(N, C, H, W, k, x.device)
    plane[:, :, top:top + H, left:left + W].copy_(x)
    y = torch.empty((N, K, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _native.check(_native.lib().tai_conv3x3_wino43_forward_blocks(
            plane.data_ptr(), k, _wino43_weights_kxk(weight, transposed).data_ptr(), bias.data_ptr(), y.data_ptr(), None, 0, 0, 0, 0,
            N, S * S * C, K, H, W, in_h, in_w, 1, 2, _ACT[act], torch.cuda.current_stream(x.device).cuda_stream),
            'tai_conv3x3_wino43_forward_blocks')
    return y


def _kxk_as_wino(x, weight, bias, act, pool, transposed=False, keep_stack=False):
    """5x5 / 7x7 "same" convolution as the Winograd 3x3 kernel over S*S shifted copies of the input (csrc/thin_conv.hip.inc,
    shift_stack): 1.56x / 1.36x fewer multiplies than the direct form MIOpen runs.  Returns y or (y, pooled).
    ``transposed``: convolve with the transposed and flipped filter (the input gradient of the same layer)."""
    N, C, H, W = x.shape
    K, k = weight.shape[1 if transposed else 0], weight.shape[2]
    S = (k + 2) // 3
    L = _native.lib()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    x = x.contiguous()
    stack = torch.empty((N, S * S * C, H + 2, W + 4), dtype=x.dtype, device=x.device)     # shifted copies, own halo
    U = _wino_weights_kxk(weight, transposed)
    y = torch.empty((N, K, H, W), dtype=x.dtype, device=x.device)
    yp = torch.empty((N, K, H // 2, W // 2), dtype=x.dtype, device=x.device) if pool else None
    with torch.cuda.device(x.device):
        _native.check(L.tai_conv_shift_stack(x.data_ptr(), stack.data_ptr(), N, C, H, W, k, stream), 'tai_conv_shift_stack')
        _native.check(L.tai_conv3x3_wino_forward_window(stack.data_ptr(), U.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                        yp.data_ptr() if pool else None, N, S * S * C, K, H, W, H + 2, W + 4,
                                                        1, 2, _ACT[act], stream), 'tai_conv3x3_wino_forward_window')
    if keep_stack:
        return y, stack
    return (y, yp) if pool else y


def _unblock3x3_weight(wb, C, k):
    """Inverse of _block3x3_weight for a gradient: [K, S*S*C, 3, 3] -> [K, C, k, k] (the taps past k are padding)."""
    K = wb.shape[0]
    S = (k + 2) // 3
    return wb.view(K, S, S, C, 3, 3).permute(0, 3, 1, 4, 2, 5).reshape(K, C, 3 * S, 3 * S)[:, :, :k, :k].contiguous()


_HALO_PLANES = {}


def halo_geometry(H, W, k):
    """The plane through which a k x k (k = 5, 7) "same" convolution over an H x W image reads its S x S blocks of 3 x 3
    taps as displaced 3 x 3 convolutions (tai_conv3x3_wino_forward_ex, shift_s): -> (S, top, left, in_h, in_w).  Image
    pixel (0, 0) sits at (top, left) = (k/2, k/2 + 1): block (0, 0)'s centre tap of output (0, 0) is then at (1, 2) -- the
    kernel's (in_oy, in_ox), in_ox even -- and block (a, b) reads 3a rows / 3b columns further, up to row H + 1 + 3 (S - 1)
    and column W + 3 + 3 (S - 1), all inside the plane; everything outside the image is zero."""
    S = (k + 2) // 3
    in_h, in_w = H + 2 + 3 * (S - 1), W + 4 + 3 * (S - 1)
    return S, k // 2, k // 2 + 1, in_h, in_w + (in_w & 1)


def halo_plane(N, C, H, W, k, device):
    """A cached zero-filled [N, C, in_h, in_w] buffer for halo_geometry(H, W, k): producers write the image into its
    interior (pooled-output window of the convolution before), the halo stays zero from the allocation on.  One buffer
    per (shape, device, stream): the consumer runs right behind the producer on that stream."""
    S, top, left, in_h, in_w = halo_geometry(H, W, k)
    key = (N, C, in_h, in_w, str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _HALO_PLANES.get(key)
    if buf is None:
        buf = torch.zeros((N, C, in_h, in_w), dtype=torch.float32, device=device)
        _HALO_PLANES[key] = buf
    return buf


def _no_grad_needed(*tensors):
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors))


def motion_enc_chain(diff, conv1, conv2, conv3):
    """MotionEnc's three convolution + ReLU + 2x2 max pool stages (mcnet.py:28-60: 5x5 1 -> g, 5x5 g -> 2g, 7x7 2g -> 4g)
    with each pooled output written straight into the halo-carrying plane the next stage reads through displaced 3 x 3
    blocks: no pooled tensor is copied, padded or stacked in between.  Returns (pooled3, [c1, c2, c3]) or None when
    the shapes do not qualify (the caller then runs the stages one by one)."""
    if not (torch.is_tensor(diff) and diff.is_cuda and diff.dtype == torch.float32 and diff.dim() == 4 and diff.shape[1] == 1):
        return None
    ws = [conv1.weight, conv2.weight, conv3.weight]
    bs = [conv1.bias, conv2.bias, conv3.bias]
    if any(b is None for b in bs) or not _no_grad_needed(diff, *ws, *bs):
        return None
    N, _, H, W = diff.shape
    g = ws[0].shape[0]
    if not (tuple(ws[0].shape) == (g, 1, 5, 5) and tuple(ws[1].shape) == (2 * g, g, 5, 5) and tuple(ws[2].shape) == (4 * g, 2 * g, 7, 7)
            and conv1.padding[0] == 2 and conv2.padding[0] == 2 and conv3.padding[0] == 3
            and H % 8 == 0 and W % 8 == 0 and g % 8 == 0 and g >= 16):
        return None
    H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
    S2, top2, left2, ih2, iw2 = halo_geometry(H2, W2, 5)
    S3, top3, left3, ih3, iw3 = halo_geometry(H4, W4, 7)
    if not (_wino_ok(N, S2 * S2 * g, 2 * g, H2, W2, 3, 3, 1) and _wino_ok(N, S3 * S3 * 2 * g, 4 * g, H4, W4, 3, 3, 1)
            and N * g * ih2 * iw2 < 2 ** 29 and N * 2 * g * ih3 * iw3 < 2 ** 29 and N * g * H * W < 2 ** 29):
        return None
    L = _native.lib()
    dev = diff.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    diff = diff.contiguous()
    c1 = torch.empty((N, g, H, W), dtype=torch.float32, device=dev)
    c2 = torch.empty((N, 2 * g, H2, W2), dtype=torch.float32, device=dev)
    c3 = torch.empty((N, 4 * g, H4, W4), dtype=torch.float32, device=dev)
    p3 = torch.empty((N, 4 * g, H4 // 2, W4 // 2), dtype=torch.float32, device=dev)
    plane2 = halo_plane(N, g, H2, W2, 5, dev)
    plane3 = halo_plane(N, 2 * g, H4, W4, 7, dev)
    U2 = None if _wino43_blocks_ok(N, g, 2 * g, H2, W2) else _wino_weights_kxk(ws[1])
    U3 = None if _wino43_blocks_ok(N, 2 * g, 4 * g, H4, W4) else _wino_weights_kxk(ws[2])
    xs2 = (ctypes.c_void_p * 1)(plane2.data_ptr())
    xs3 = (ctypes.c_void_p * 1)(plane3.data_ptr())
    with torch.cuda.device(dev):
        _native.check(L.tai_conv_cin1_forward_maxpool_window(diff.data_ptr(), ws[0].contiguous().data_ptr(), bs[0].data_ptr(),
                                                            c1.data_ptr(), plane2.data_ptr(), N, g, H, W, 5, 1, ih2, iw2, top2, left2,
                                                            stream), 'tai_conv_cin1_forward_maxpool_window')
        if _wino43_blocks_ok(N, g, 2 * g, H2, W2):               # the 4 x 4 tile (set_winograd_tile): 1.78x fewer MFMAs
            _native.check(L.tai_conv3x3_wino43_forward_blocks(plane2.data_ptr(), 5, _wino43_weights_kxk(ws[1]).data_ptr(), bs[1].data_ptr(),
                                                             c2.data_ptr(), plane3.data_ptr(), ih3, iw3, top3, left3, N, S2 * S2 * g, 2 * g,
                                                             H2, W2, ih2, iw2, 1, 2, 1, stream), 'tai_conv3x3_wino43_forward_blocks')
        else:
            _native.check(L.tai_conv3x3_wino_forward_ex(xs2, 1, 5, U2.data_ptr(), bs[1].data_ptr(), c2.data_ptr(), plane3.data_ptr(),
                                                       ih3, iw3, top3, left3, None, None, N, S2 * S2 * g, 2 * g, H2, W2, ih2, iw2, 1, 2, 1,
                                                       stream), 'tai_conv3x3_wino_forward_ex')
        if _wino43_blocks_ok(N, 2 * g, 4 * g, H4, W4):
            _native.check(L.tai_conv3x3_wino43_forward_blocks(plane3.data_ptr(), 7, _wino43_weights_kxk(ws[2]).data_ptr(), bs[2].data_ptr(),
                                                             c3.data_ptr(), p3.data_ptr(), 0, 0, 0, 0, N, S3 * S3 * 2 * g, 4 * g,
                                                             H4, W4, ih3, iw3, 1, 2, 1, stream), 'tai_conv3x3_wino43_forward_blocks')
        else:
            _native.check(L.tai_conv3x3_wino_forward_ex(xs3, 1, 7, U3.data_ptr(), bs[2].data_ptr(), c3.data_ptr(), p3.data_ptr(),
                                                       0, 0, 0, 0, None, None, N, S3 * S3 * 2 * g, 4 * g, H4, W4, ih3, iw3, 1, 2, 1,
                                                       stream), 'tai_conv3x3_wino_forward_ex')
    return p3, [c1, c2, c3]


def conv_bias_unpool_add(x, weight, bias, padding, addx, keep_plain=True):
    """(y, y + fixed_unpooling(addx)) with y = conv(x) + bias, no activation: the last convolution of a Residual block
    (mcnet.py:172-176) and DecCnn's unpool + residual add (mcnet.py:234-236) -- a Winograd tile is one unpooling cell, so
    the sum is a second output of the convolution's epilogue.  ``x`` may be a tuple of cat operands.  ``keep_plain=False``:
    only the sum is produced (returns (None, sum)): nothing reads the full-resolution residual itself."""
    parts = list(x) if isinstance(x, (list, tuple)) else [x]
    x0 = parts[0]
    Co, Ci, kh, kw = weight.shape
    N, Cp, H, W = x0.shape
    fused = (x0.is_cuda and x0.dtype == torch.float32 and bias is not None and addx.is_cuda and addx.dtype == torch.float32
             and tuple(addx.shape) == (N, Co, H // 2, W // 2) and len(parts) <= 4 and Cp * len(parts) == Ci
             and (len(parts) == 1 or Cp % 8 == 0) and all(q.shape == x0.shape and q.dtype == x0.dtype for q in parts)
             and _no_grad_needed(weight, bias, addx, *parts) and _wino_ok(N, Ci, Co, H, W, kh, kw, padding)
             and N * Co * H * W < 2 ** 29)
    if not fused:
        from .mcnet import unpool2x_add
        y = conv_bias_act(x, weight, bias, padding, None)
        return (y if keep_plain else None), unpool2x_add(addx, y)
    L = _native.lib()
    parts = [q.contiguous() for q in parts]
    addx = addx.contiguous()
    y = torch.empty((N, Co, H, W), dtype=torch.float32, device=x0.device)
    y2 = torch.empty_like(y) if keep_plain else None
    ptrs = (ctypes.c_void_p * len(parts))(*[q.data_ptr() for q in parts])
    if _wino43_ok(N, Ci, Co, H, W, len(parts), weight):      # wide layer: F(4x4, 3x3) (set_winograd_tile): a tile holds four unpooling cells
        U = _wino43_weights(weight, False)
        with torch.cuda.device(x0.device):
            _native.check(L.tai_conv3x3_wino43_forward_ex(ptrs, len(parts), U.data_ptr(), bias.data_ptr(), y.data_ptr(), None, addx.data_ptr(),
                                                          y2.data_ptr() if keep_plain else None, N, Ci, Co, H, W, 0,
                                                          torch.cuda.current_stream(x0.device).cuda_stream), 'tai_conv3x3_wino43_forward_ex')
        return (y, y2) if keep_plain else (None, y)
    U = _wino_weights(weight, False)
    with torch.cuda.device(x0.device):
        _native.check(L.tai_conv3x3_wino_forward_ex(ptrs, len(parts), 0, U.data_ptr(), bias.data_ptr(), y.data_ptr(), None, 0, 0, 0, 0,
                                                   addx.data_ptr(), y2.data_ptr() if keep_plain else None, N, Ci, Co, H, W, H, W, 0, 0, 0,
                                                   torch.cuda.current_stream(x0.device).cuda_stream), 'tai_conv3x3_wino_forward_ex')
    return (y, y2) if keep_plain else (None, y)


def _kxk_ok(N, Ci, Co, H, W, kh, kw, padding):
    S = (kh + 2) // 3
    return (kh == kw and kh in (5, 7) and padding == kh // 2 and W % 4 == 0 and Ci >= 16
            and N * S * S * Ci * (H + 2) * (W + 4) < 2 ** 29 and _wino_ok(N, S * S * Ci, Co, H, W, 3, 3, 1))


# Below this the kernel leaves most of the 256 CUs idle and MIOpen is as fast (tools/conv_path_times.py).  72 rather than round
# 1's 96: configs[1]'s last MIOpen layers (the kernel network's 512 -> 512 at 4 x 4 over 160 samples: 80 workgroups) run 132 us here
# against 121-155 us there, and MIOpen's implicit-GEMM kernel sums with atomics -- with it the forward's kernel-network outputs
# differed by 1.5e-6 from replay to replay; without it 1,500 replays of the whole forward are bit-identical (tools/soak_forward.py).
WINO_MIN_WORKGROUPS = 72

# Training: convolutions of a tuple of channel parts go through _WinoConv3x3Parts (no torch.cat in the forward, one contiguous input
# gradient per part); False restores the concatenating path (A/B, tests)
PARTS_UNDER_AUTOGRAD = True


def _wino_ok(N, Ci, Co, H, W, kh, kw, padding, min_ci=8):
    """``min_ci``: the kernel pads the input channels to a multiple of 8 with zero weights, so fewer than 8 work -- at the cost of a
    whole chunk; the inference paths take that down to 2 (TAI_color's first ContentEnc layer, 3 -> 64 at 256 x 256: 276 us against
    MIOpen's 620), the training path does not (its input gradient would run 64 rows of MFMA for 3 output channels)."""
    return (kh == kw == 3 and padding == 1 and H % 2 == 0 and W % 2 == 0 and Ci >= min_ci and N * max(Ci, Co) * H * W < 2 ** 29
            and ((N * (H // 2) * (W // 2) + 63) // 64) * ((Co + 63) // 64) >= WINO_MIN_WORKGROUPS)


def _usable_out(out, shape, like):
    return (out is not None and tuple(out.shape) == tuple(shape) and out.is_contiguous() and out.dtype == like.dtype
            and out.device == like.device)


def _wino_launch(x, U, bias, N, Ci, Co, H, W, act):
    y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _native.check(_native.lib().tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Ci, Co,
                                                            H, W, _ACT[act], torch.cuda.current_stream(x.device).cuda_stream),
                      'tai_conv3x3_wino_forward')
    return y


# Under autograd the forward and the input gradient (the same convolution with the weight transposed and flipped) take the 4 x 4 tile on
# the layers the inference path gives it: the training forward then runs the arithmetic of the inference forward (False: F(2x2, 3x3)
# for everything under autograd, as in rounds 2-4)
WINO43_UNDER_AUTOGRAD = True


def _conv3x3_autograd_launch(x, weight, eff_transposed, bias, N, Ci, Co, H, W, act):
    """conv(x, w_eff) with w_eff = the weight in orientation ``eff_transposed``: F(4x4, 3x3) where _wino43_ok says so, else F(2x2, 3x3)."""
    if WINO43_UNDER_AUTOGRAD and _wino43_ok(N, Ci, Co, H, W, 1, weight):
        y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().tai_conv3x3_wino43_forward(x.data_ptr(), _wino43_weights(weight, eff_transposed).data_ptr(), bias.data_ptr(),
                                                                  y.data_ptr(), N, Ci, Co, H, W, _ACT[act],
                                                                  torch.cuda.current_stream(x.device).cuda_stream), 'tai_conv3x3_wino43_forward')
        return y
    return _wino_launch(x, _wino_weights(weight, eff_transposed), bias, N, Ci, Co, H, W, act)


def wino_conv3x3_plain(x, weight, transposed=False):
    """conv2d(x, w_eff, padding=1) without bias, activation or autograd on the in-tree Winograd kernels (F(4x4, 3x3) / F(2x2, 3x3) as
    _conv3x3_autograd_launch chooses); ``transposed``: w_eff = the weight transposed and flipped (the input-gradient convolution of the
    same weight).  None where neither kernel takes the shape -- the caller then keeps its own route.  (The discriminator's 4x4 stride-2
    layers as 3x3 layers on space-to-depth planes: sn_discriminator.py.)"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32 and x.dim() == 4):
        return None
    N, Ci, H, W = x.shape
    Co = weight.shape[1] if transposed else weight.shape[0]
    if (weight.shape[0] if transposed else weight.shape[1]) != Ci or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError('wino_conv3x3_plain: weight %s does not fit input %s' % (tuple(weight.shape), tuple(x.shape)))
    if not ((WINO43_UNDER_AUTOGRAD and _wino43_ok(N, Ci, Co, H, W, 1, weight)) or _wino_ok(N, Ci, Co, H, W, 3, 3, 1)):
        return None
    return _conv3x3_autograd_launch(x, weight, transposed, _zero_bias(Co, x.device), N, Ci, Co, H, W, None)


_ZERO_BIAS = {}


def _zero_bias(n, device):
    """A cached all-zero fp32 bias (read-only) for the input-gradient convolutions: one fill kernel per call otherwise."""
    key = (n, device)
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = torch.zeros(n, dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():       # (a fill recorded into a graph has not run yet: do not keep it)
            _ZERO_BIAS[key] = z
    return z


class _GrowingWorkspace(object):
    """Per-device fp32 scratch that grows to the largest request.  A captured graph (environments.train_step with
    graph_step) bakes the buffer's raw pointer into its kernel nodes, so once any request has been served under capture a
    superseded buffer is retired to a keep-alive list instead of being released: a later, larger eager request must not
    hand the old block back to the caching allocator while replays still write partial sums into it."""

    def __init__(self):
        self.current, self.retired, self.captured = {}, [], False

    def get(self, device, floats):
        self.captured = self.captured or torch.cuda.is_current_stream_capturing()
        ws = self.current.get(device)
        if ws is None or ws.numel() < floats:
            if ws is not None and self.captured:
                self.retired.append(ws)
            ws = self.current[device] = torch.empty(floats, dtype=torch.float32, device=device)
        return ws


_WRW_WORKSPACE = _GrowingWorkspace()


def wino_weight_grad(x, grad_out, with_bias=False, window=None):
    """dL/dw [Co, Ci, 3, 3] of y = conv2d(x, w, padding=1) from x [N, Ci, H, W] and dL/dy [N, Co, H, W], by
    ``tai_conv3x3_wino_wrw`` (Winograd-domain weight gradient on the fp32 MFMA pipe); None if the shape is not supported
    (odd H, W above 16 and not a multiple of 16, a tensor of 2 GiB or more; rows of fewer than 16 pixels are widened with zeros).  ``with_bias``: returns (dw, dbias), the bias gradient
    summed by the same kernel.  ``window`` = (in_oy, in_ox): x is a plane [N, Ci, in_h, in_w] that carries its own halo,
    with the pixel under output (0, 0) at (in_oy, in_ox) (the shifted-copy stack of the 5x5 / 7x7 layers).  The workspace
    (partial sums per workgroup) is kept per device and grows to the largest request."""
    N, Ci = x.shape[0], x.shape[1]
    Co, H, W = grad_out.shape[1], grad_out.shape[2], grad_out.shape[3]
    L = _native.lib()
    if window is None and W % 16 and W < 16 and H % 2 == 0 and x.dtype == torch.float32 and grad_out.dtype == torch.float32:
        # rows shorter than the kernel's 16 pixels (the 8 x 8 and 4 x 4 layers of the kernel network and of the discriminator): both planes
        # widened to 16 columns with zeros -- the added output-gradient columns contribute nothing, the added input columns are the zero
        # padding the last real column sees anyway -- instead of ATen's weight gradient, whose summation order changes from run to run
        x, grad_out, W = F.pad(x, (0, 16 - W)), F.pad(grad_out, (0, 16 - W)), 16
    floats = L.tai_conv3x3_wino_wrw_workspace_floats(N, Ci, Co, H, W)
    if floats < 0 or N * Ci * x.shape[2] * x.shape[3] >= 2 ** 29:
        return None
    ws = _WRW_WORKSPACE.get(x.device, floats)
    dw = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty(Co, dtype=torch.float32, device=x.device) if with_bias else None
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if window is None:
            _native.check(L.tai_conv3x3_wino_wrw(x.data_ptr(), grad_out.data_ptr(), dw.data_ptr(), db.data_ptr() if with_bias else None,
                                                 ws.data_ptr(), N, Ci, Co, H, W, stream), 'tai_conv3x3_wino_wrw')
        else:
            _native.check(L.tai_conv3x3_wino_wrw_window(x.data_ptr(), grad_out.data_ptr(), dw.data_ptr(),
                                                        db.data_ptr() if with_bias else None, ws.data_ptr(), N, Ci, Co, H, W,
                                                        x.shape[2], x.shape[3], window[0], window[1], stream),
                          'tai_conv3x3_wino_wrw_window')
    return (dw, db) if with_bias else dw


class _WinoConv3x3(torch.autograd.Function):
    """Training form of the 3x3 convolution: the forward and the input gradient (the same convolution with the weight
    transposed and flipped) run on the Winograd-MFMA kernel, the weight gradient on its Winograd-domain counterpart
    (wino_weight_grad; MIOpen's aten.convolution_backward for shapes that one does not take).  ``transposed``: the weight is a ConvTranspose2d(k 3, stride 1, padding 1) weight."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, transposed):
        x = x.contiguous()
        Co, Ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        N, _, H, W = x.shape
        y = _conv3x3_autograd_launch(x, weight, transposed, bias, N, Ci, Co, H, W, act)
        ctx.act, ctx.transposed = act, transposed
        ctx.save_for_backward(x, weight, y if act is not None else None)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, y = ctx.saved_tensors
        g = grad_out.contiguous()
        if ctx.act == 'relu':
            g = torch.ops.aten.threshold_backward(g, y, 0)        # g where y > 0, else 0: one kernel
        elif ctx.act == 'tanh':
            g = torch.ops.aten.tanh_backward(g, y)                  # g * (1 - y^2): one kernel
        N, Ci, H, W = x.shape
        Co = g.shape[1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # d/dx of conv(x, w_eff) is conv(g, w_eff transposed and flipped): the other orientation of the same weight
            zero = _zero_bias(Ci, g.device)
            gx = _conv3x3_autograd_launch(g, weight, not ctx.transposed, zero, N, Co, Ci, H, W, None)
        if ctx.needs_input_grad[1]:
            gw_eff = None
            if g.dtype == torch.float32 and x.dtype == torch.float32:
                if ctx.needs_input_grad[2]:
                    both = wino_weight_grad(x, g, with_bias=True)
                    if both is not None:
                        gw_eff, gb = both
                else:
                    gw_eff = wino_weight_grad(x, g)
            if gw_eff is None:                                    # shapes the Winograd weight-gradient kernel does not take
                w_eff = _as_conv_weight(weight, ctx.transposed)
                gw_eff = torch.ops.aten.convolution_backward(g, x, w_eff, [Co], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                             [False, True, False])[1]
            gw = _as_conv_weight(gw_eff, ctx.transposed)          # the transpose-and-flip is its own inverse
        if ctx.needs_input_grad[2] and gb is None:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, None, None


class _WinoConv3x3Parts(torch.autograd.Function):
    """``_WinoConv3x3`` on an input given as up to four equal channel parts (the operands of a ``torch.cat`` along the channels that is
    never materialised: Residual, CombLayers, the ConvLSTM's (input, h), the kernel network's 1024-channel input).  Under autograd the
    parts used to be concatenated first (one copy of the layer's whole input per call) and the input gradient came back as ONE tensor,
    whose channel slices every producer then had to copy into contiguous memory for its own backward.  Here the forward reads the parts
    where they lie (``tai_conv3x3_wino_forward_parts``: the same bits as the convolution of the concatenation), the input gradient of
    part i is its own launch of the Winograd kernel over the K-slice of the transposed, flipped weight that belongs to it (the layer's
    work split by output channels: no flop more, a contiguous result per part) and the weight gradient is taken per part and joined
    along the input channels (a [Co, Ci, 3, 3] copy)."""

    @staticmethod
    def forward(ctx, weight, bias, act, transposed, *parts):
        x0 = parts[0]
        Co, Ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        N, Cp, H, W = x0.shape
        L = _native.lib()
        y = torch.empty((N, Co, H, W), dtype=x0.dtype, device=x0.device)
        ptrs = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        with torch.cuda.device(x0.device):
            if WINO43_UNDER_AUTOGRAD and _wino43_ok(N, Ci, Co, H, W, len(parts), weight):
                _native.check(L.tai_conv3x3_wino43_forward_parts(ptrs, len(parts), _wino43_weights(weight, transposed).data_ptr(), bias.data_ptr(),
                                                                 y.data_ptr(), N, Ci, Co, H, W, _ACT[act],
                                                                 torch.cuda.current_stream(x0.device).cuda_stream), 'tai_conv3x3_wino43_forward_parts')
            else:
                _native.check(L.tai_conv3x3_wino_forward_parts(ptrs, len(parts), _wino_weights(weight, transposed).data_ptr(), bias.data_ptr(),
                                                               y.data_ptr(), N, Ci, Co, H, W, _ACT[act],
                                                               torch.cuda.current_stream(x0.device).cuda_stream), 'tai_conv3x3_wino_forward_parts')
        ctx.act, ctx.transposed, ctx.nparts = act, transposed, len(parts)
        ctx.save_for_backward(weight, y if act is not None else None, *parts)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        weight, y, parts = saved[0], saved[1], saved[2:]
        g = grad_out.contiguous()
        if ctx.act == 'relu':
            g = torch.ops.aten.threshold_backward(g, y, 0)
        elif ctx.act == 'tanh':
            g = torch.ops.aten.tanh_backward(g, y)
        N, Cp, H, W = parts[0].shape
        Co, n = g.shape[1], ctx.nparts
        gparts = [None] * n
        for i in range(n):
            if ctx.needs_input_grad[4 + i]:
                if WINO43_UNDER_AUTOGRAD and _wino43_ok(N, Co, Cp, H, W, 1, weight):
                    gparts[i] = torch.empty((N, Cp, H, W), dtype=g.dtype, device=g.device)
                    with torch.cuda.device(g.device):
                        _native.check(_native.lib().tai_conv3x3_wino43_forward(
                            g.data_ptr(), _wino_weights_input_grad_part(weight, ctx.transposed, i, n, tile=4).data_ptr(),
                            _zero_bias(Cp, g.device).data_ptr(), gparts[i].data_ptr(), N, Co, Cp, H, W, 0,
                            torch.cuda.current_stream(g.device).cuda_stream), 'tai_conv3x3_wino43_forward')
                else:
                    gparts[i] = _wino_launch(g, _wino_weights_input_grad_part(weight, ctx.transposed, i, n), _zero_bias(Cp, g.device),
                                             N, Co, Cp, H, W, None)
        gw = gb = None
        if ctx.needs_input_grad[0]:
            want_bias = ctx.needs_input_grad[1]
            pieces = []
            for i in range(n):
                r = wino_weight_grad(parts[i], g, with_bias=(want_bias and i == 0))
                if r is None:
                    pieces = None
                    break
                if want_bias and i == 0:
                    r, gb = r
                pieces.append(r)
            if pieces is None:                                        # shapes the Winograd weight-gradient kernel does not take
                w_eff = _as_conv_weight(weight, ctx.transposed)
                gw_eff = torch.ops.aten.convolution_backward(g, torch.cat(parts, dim=1), w_eff, [Co], [1, 1], [1, 1], [1, 1], False,
                                                             [0, 0], 1, [False, True, False])[1]
            else:
                gw_eff = torch.cat(pieces, dim=1)
            gw = _as_conv_weight(gw_eff, ctx.transposed)
        if ctx.needs_input_grad[1] and gb is None:
            gb = g.sum((0, 2, 3))
        return (gw, gb, None, None) + tuple(gparts)


def _wino_weights_input_grad_part(weight, transposed, i, nparts, tile=2):
    """Transformed weights of the input-gradient convolution of channel part ``i`` of ``nparts``: conv(g, w_i transposed and flipped)
    with w_i the slice of the layer's effective weight [Co, Ci, 3, 3] over that part's input channels; ``tile`` 2 / 4: the F(2x2, 3x3) /
    F(4x4, 3x3) layout."""
    def make():
        w_eff = _as_conv_weight(weight.detach(), transposed)                              # [Co, Ci, 3, 3]
        Cp = w_eff.shape[1] // nparts
        w = w_eff[:, i * Cp:(i + 1) * Cp].transpose(0, 1).flip(2, 3).contiguous()          # [Cp, Co, 3, 3]
        K, C = w.shape[0], w.shape[1]
        L = _native.lib()
        pre = 'tai_conv3x3_wino43' if tile == 4 else 'tai_conv3x3_wino'
        U = torch.empty(getattr(L, pre + '_weight_floats')(K, C), dtype=torch.float32, device=w.device)
        if tile == 2:
            _registered(U)
        with torch.cuda.device(w.device):
            _native.check(getattr(L, pre + '_transform_weights')(w.data_ptr(), U.data_ptr(), K, C, torch.cuda.current_stream(w.device).cuda_stream),
                          pre + '_transform_weights')
        return U
    return _cached(weight, ('wino_input_grad_part', transposed, i, nparts, tile, _WINO_ARITH[0]), make)


class _WinoConvKxK(torch.autograd.Function):
    """Training form of the 5x5 / 7x7 convolutions (MotionEnc, mcnet.py:36-47): forward and input gradient as 3x3 blocks on
    the Winograd-MFMA kernel (_kxk_as_wino), weight and bias gradients on the Winograd-domain weight-gradient kernel over the
    same stack of shifted copies (MIOpen's where that kernel does not take the shape)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        ctx.act = act
        if WINO43_UNDER_AUTOGRAD and act in (None, 'relu') and _wino43_blocks_ok(N, Ci, Co, H, W):
            # the 4 x 4 tile over a halo plane; the stack of shifted copies the weight gradient reads is built in the backward pass
            # (134 / 151 MB per MotionEnc call no longer kept from forward to backward)
            y = _kxk_as_blocks(x, weight, bias, act)
            ctx.save_for_backward(x, weight, y if act is not None else None)
            return y
        y, stack = _kxk_as_wino(x, weight, bias, act, False, keep_stack=True)
        # the stack of shifted copies (S*S times the input) is kept for the weight gradient: 134 / 151 MB per MotionEnc call
        ctx.save_for_backward(x, weight, y if act is not None else None, stack)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        x, weight, y = saved[0], saved[1], saved[2]
        g = grad_out.contiguous()
        if ctx.act == 'relu':
            g = torch.ops.aten.threshold_backward(g, y, 0)
        elif ctx.act == 'tanh':
            g = torch.ops.aten.tanh_backward(g, y)                  # g * (1 - y^2): one kernel
        Co, Ci, k = weight.shape[0], weight.shape[1], weight.shape[2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if WINO43_UNDER_AUTOGRAD and _wino43_blocks_ok(g.shape[0], Co, Ci, g.shape[2], g.shape[3]):
                gx = _kxk_as_blocks(g, weight, _zero_bias(Ci, g.device), None, transposed=True)
            else:
                gx = _kxk_as_wino(g, weight, _zero_bias(Ci, g.device), None, False, transposed=True)
        if ctx.needs_input_grad[1]:
            if len(saved) > 3:
                stack = saved[3]
            else:
                N, _, H, W = x.shape
                S = (k + 2) // 3
                stack = torch.empty((N, S * S * Ci, H + 2, W + 4), dtype=x.dtype, device=x.device)
                with torch.cuda.device(x.device):
                    _native.check(_native.lib().tai_conv_shift_stack(x.data_ptr(), stack.data_ptr(), N, Ci, H, W, k,
                                                                     torch.cuda.current_stream(x.device).cuda_stream), 'tai_conv_shift_stack')
            # the weight gradient of the blocked 3x3 form over the stack (it carries its halo: origin (1, 2)), then un-blocked
            both = wino_weight_grad(stack, g, with_bias=True, window=(1, 2))
            if both is not None:
                gw, gb = _unblock3x3_weight(both[0], Ci, k), both[1]
            else:
                gw = torch.ops.aten.convolution_backward(g, x, weight, [Co], [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        if ctx.needs_input_grad[2] and gb is None:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, None


_THIN_WORKSPACE = _GrowingWorkspace()


def thin_weight_grad(big, thin, k):
    """(dw [Cb, k, k], db [Cb]) with dw[cb][a][b] = sum of big[n, cb, y, x] * thin[n, 0, y + a - k/2, x + b - k/2] and
    db[cb] = sum of big[n, cb]: one pass over ``big`` (tai_thin_conv_wrw)."""
    N, Cb, H, W = big.shape
    ws = _THIN_WORKSPACE.get(big.device, N * Cb * 32)
    dw = torch.empty((Cb, k, k), dtype=torch.float32, device=big.device)
    db = torch.empty(Cb, dtype=torch.float32, device=big.device)
    with torch.cuda.device(big.device):
        _native.check(_native.lib().tai_thin_conv_wrw(big.data_ptr(), thin.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                      N, Cb, H, W, k, torch.cuda.current_stream(big.device).cuda_stream),
                      'tai_thin_conv_wrw')
    return dw, db


class _ThinInConv(torch.autograd.Function):
    """Training form of the one-input-channel layers (nn.Conv2d(1, gf, k) + ReLU, mcnet.py:28-31, 79-81): forward on
    tai_conv_cin1_forward, weight and bias gradients in one pass over the output gradient (thin_weight_grad); the input is
    a frame (difference), which needs no gradient -- if it does, ATen's."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        N, _, H, W = x.shape
        Co, k = weight.shape[0], weight.shape[2]
        y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().tai_conv_cin1_forward(x.data_ptr(), weight.detach().contiguous().data_ptr(), bias.data_ptr(),
                                                              y.data_ptr(), N, Co, H, W, k, _ACT[act],
                                                              torch.cuda.current_stream(x.device).cuda_stream), 'tai_conv_cin1_forward')
        ctx.act = act
        ctx.save_for_backward(x, weight, y if act is not None else None)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, y = ctx.saved_tensors
        g = grad_out.contiguous()
        if ctx.act == 'relu':
            g = torch.ops.aten.threshold_backward(g, y, 0)
        Co, k = weight.shape[0], weight.shape[2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # (the frame difference of an autoregressive step is itself generated)
            N, _, H, W = x.shape
            L = _native.lib()
            stream = torch.cuda.current_stream(x.device).cuda_stream
            wf = weight.detach().flip(2, 3).transpose(0, 1).contiguous()                 # [1, Co, k, k]: the filter flipped
            gx = torch.empty_like(x)
            with torch.cuda.device(x.device):
                if k == 5:
                    _native.check(L.tai_conv_cout1_5x5_forward(g.data_ptr(), wf.data_ptr(), None, gx.data_ptr(), N, Co, H, W, stream),
                                  'tai_conv_cout1_5x5_forward')
                else:
                    _native.check(L.tai_conv_cout1_3x3_forward(g.data_ptr(), wf.data_ptr(), _zero_bias(1, x.device).data_ptr(),
                                                               gx.data_ptr(), N, Co, H, W, 0, stream), 'tai_conv_cout1_3x3_forward')
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, gb = thin_weight_grad(g, x, k)
            gw = dw.view_as(weight)
        return gx, gw, gb, None


class _ThinOutConv(torch.autograd.Function):
    """Training form of the one-output-channel layer (nn.ConvTranspose2d(gf, 1, 3, padding=1) + Tanh, mcnet.py:223-224):
    forward on tai_conv_cout1_3x3_forward; the input gradient is a one-input-channel convolution of the output gradient
    with the flipped filter (tai_conv_cin1_forward), the weight gradient one pass over the input (thin_weight_grad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, transposed):
        x = x.contiguous()
        N, Ci, H, W = x.shape
        wd = _cached(weight, ('direct', transposed), lambda: _as_conv_weight(weight.detach(), transposed).contiguous())   # [1, Ci, 3, 3]
        y = torch.empty((N, 1, H, W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().tai_conv_cout1_3x3_forward(x.data_ptr(), wd.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Ci,
                                                                   H, W, _ACT[act], torch.cuda.current_stream(x.device).cuda_stream),
                          'tai_conv_cout1_3x3_forward')
        ctx.act, ctx.transposed = act, transposed
        ctx.save_for_backward(x, weight, y if act is not None else None)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, y = ctx.saved_tensors
        g = grad_out.contiguous()
        if ctx.act == 'relu':
            g = torch.ops.aten.threshold_backward(g, y, 0)
        elif ctx.act == 'tanh':
            g = torch.ops.aten.tanh_backward(g, y)                  # g * (1 - y^2): one kernel
        N, Ci, H, W = x.shape
        wd = _cached(weight, ('direct', ctx.transposed), lambda: _as_conv_weight(weight.detach(), ctx.transposed).contiguous())
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # d/dx_c of sum_c conv(x_c, wd[0, c]) is conv(g, wd[0, c] flipped): Ci output channels from the one-channel g
            wf = wd.flip(2, 3).transpose(0, 1).contiguous()                           # [Ci, 1, 3, 3]
            gx = torch.empty_like(x)
            zero = _zero_bias(Ci, x.device)
            with torch.cuda.device(x.device):
                _native.check(_native.lib().tai_conv_cin1_forward(g.data_ptr(), wf.data_ptr(), zero.data_ptr(), gx.data_ptr(), N, Ci, H,
                                                                  W, 3, 0, torch.cuda.current_stream(x.device).cuda_stream),
                              'tai_conv_cin1_forward')
        if ctx.needs_input_grad[1]:
            # dwd[0][c][a][b] = sum g[y, x] x_c[y + a - 1, x + b - 1] = thin_weight_grad(x, g)[c][2 - a][2 - b]
            dwd = thin_weight_grad(x, g, 3)[0].flip(1, 2).unsqueeze(0)               # [1, Ci, 3, 3] in conv2d layout
            gw = _as_conv_weight(dwd, ctx.transposed).contiguous().view_as(weight)
        if ctx.needs_input_grad[2]:
            gb = g.sum().view(1)
        return gx, gw, gb, None, None


class _ActPool2x2(torch.autograd.Function):
    """(y, max_pool2d(y, 2)) with y = relu(z) or z, under autograd: one pass forward, and ONE pass backward for what ATen
    runs as max_pool2d backward + the sum of the two gradient paths into y + threshold_backward."""

    @staticmethod
    def forward(ctx, z, relu):
        z = z.contiguous()
        N, C, H, W = z.shape
        y = torch.empty_like(z)
        yp = torch.empty((N, C, H // 2, W // 2), dtype=z.dtype, device=z.device)
        with torch.cuda.device(z.device):
            _native.check(_native.lib().tai_act_maxpool2x2_forward(z.data_ptr(), y.data_ptr(), yp.data_ptr(), N * C, H, W, int(relu),
                                                                   torch.cuda.current_stream(z.device).cuda_stream),
                          'tai_act_maxpool2x2_forward')
        ctx.relu = bool(relu)
        ctx.save_for_backward(y)
        return y, yp

    @staticmethod
    def backward(ctx, gy, gyp):
        y, = ctx.saved_tensors
        N, C, H, W = y.shape
        gy = gy.contiguous() if gy is not None else None
        gyp = gyp.contiguous() if gyp is not None else None
        gz = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _native.check(_native.lib().tai_act_maxpool2x2_backward(gy.data_ptr() if gy is not None else None,
                                                                    gyp.data_ptr() if gyp is not None else None, y.data_ptr(),
                                                                    gz.data_ptr(), N * C, H, W, int(ctx.relu),
                                                                    torch.cuda.current_stream(y.device).cuda_stream),
                          'tai_act_maxpool2x2_backward')
        return gz, None


def conv_bias_act_maxpool(x, weight, bias, padding, act):
    """(y, max_pool2d(y, 2)) with y = conv_bias_act(x, ...): the kernels that own a whole 2x2 window per lane (Winograd,
    one-input-channel) write the pooled tensor in their epilogue instead of leaving a second pass over y to ATen."""
    Co, Ci, kh, kw = weight.shape
    fused = (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and bias is not None and act in (None, 'relu')
             and x.shape[2] % 2 == 0 and x.shape[3] % 4 == 0
             and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad)))
    if fused:
        N, _, H, W = x.shape
        L = _native.lib()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        thin_in = Ci == 1 and kh == kw and kh in (3, 5) and padding == kh // 2 and Co >= 16
        if not thin_in and _kxk_ok(N, Ci, Co, H, W, kh, kw, padding):
            return _kxk_as_wino(x, weight, bias, act, True)
        if thin_in or _wino_ok(N, Ci, Co, H, W, kh, kw, padding):
            x = x.contiguous()
            y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
            yp = torch.empty((N, Co, H // 2, W // 2), dtype=x.dtype, device=x.device)
            with torch.cuda.device(x.device):
                if thin_in:
                    _native.check(L.tai_conv_cin1_forward_maxpool(x.data_ptr(), weight.contiguous().data_ptr(), bias.data_ptr(),
                                                                  y.data_ptr(), yp.data_ptr(), N, Co, H, W, kh, _ACT[act], stream),
                                  'tai_conv_cin1_forward_maxpool')
                elif kh == kw == 3 and padding == 1 and _wino43_ok(N, Ci, Co, H, W, 1, weight):      # wide layer: F(4x4, 3x3): a tile is four pooling windows
                    U = _wino43_weights(weight, False)
                    xs = (ctypes.c_void_p * 1)(x.data_ptr())
                    _native.check(L.tai_conv3x3_wino43_forward_ex(xs, 1, U.data_ptr(), bias.data_ptr(), y.data_ptr(), yp.data_ptr(), None, None,
                                                                  N, Ci, Co, H, W, _ACT[act], stream), 'tai_conv3x3_wino43_forward_ex')
                else:
                    U = _wino_weights(weight, False)
                    _native.check(L.tai_conv3x3_wino_forward_maxpool(x.data_ptr(), U.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                                     yp.data_ptr(), N, Ci, Co, H, W, _ACT[act], stream),
                                  'tai_conv3x3_wino_forward_maxpool')
            return y, yp
    if (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and act in (None, 'relu') and x.shape[2] % 2 == 0
            and x.shape[3] % 4 == 0 and torch.is_grad_enabled()):
        # training: the convolution without its activation, then activation + pool as one Function (one backward pass for
        # the pool's scatter, the sum of the two gradient paths into y and the ReLU mask)
        return _ActPool2x2.apply(conv_bias_act(x, weight, bias, padding, None), act == 'relu')
    y = conv_bias_act(x, weight, bias, padding, act)
    return y, F.max_pool2d(y, 2)


def conv_bias_act(x, weight, bias, padding, act, transposed=False, out=None):
    """act(conv2d(x, weight, stride 1, padding) + bias), act in {None, 'relu', 'tanh'}.  ``transposed``: ``weight`` is
    the [in, out, 3, 3] weight of a ConvTranspose2d(k 3, stride 1, padding 1), which is the same convolution with the
    weight transposed and flipped.  ``x`` may be a list of up to four tensors, meaning their concatenation along the
    channels; the Winograd kernel reads the parts where they lie, every other path concatenates them first.
    ``out``: a contiguous tensor (e.g. a batch slice of a larger buffer) to receive the result; the Winograd paths write it
    directly, every other path copies into it."""
    if out is not None:
        y = _conv_bias_act(x, weight, bias, padding, act, transposed, out)
        if y is not out:
            out.copy_(y)
        return out
    return _conv_bias_act(x, weight, bias, padding, act, transposed, None)


def _conv_bias_act(x, weight, bias, padding, act, transposed, out):
    if isinstance(x, (list, tuple)):
        parts = list(x)
        x0 = parts[0]
        kh, kw = weight.shape[2], weight.shape[3]
        Co, Ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        N, Cp, H, W = x0.shape
        direct = (len(parts) <= 4 and x0.is_cuda and x0.dtype == torch.float32 and bias is not None and Cp % 8 == 0
                  and Cp * len(parts) == Ci and all(p.shape == x0.shape and p.is_contiguous() and p.dtype == x0.dtype for p in parts)
                  and not (torch.is_grad_enabled() and (weight.requires_grad or bias.requires_grad or any(p.requires_grad for p in parts)))
                  and _wino_ok(N, Ci, Co, H, W, kh, kw, padding))
        if not direct:
            if (len(parts) <= 4 and x0.is_cuda and x0.dtype == torch.float32 and bias is not None and Cp % 8 == 0 and Cp * len(parts) == Ci
                    and kh == kw == 3 and all(p.shape == x0.shape and p.is_contiguous() and p.dtype == x0.dtype for p in parts)
                    and torch.is_grad_enabled() and _wino_ok(N, Ci, Co, H, W, 3, 3, padding) and _wino_ok(N, Co, Cp, H, W, 3, 3, padding)
                    and PARTS_UNDER_AUTOGRAD):
                return _WinoConv3x3Parts.apply(weight, bias, act, transposed, *parts)     # training: the parts are read where they lie
            return _conv_bias_act(torch.cat(parts, dim=1), weight, bias, padding, act, transposed, out)
        L = _native.lib()
        y = out if _usable_out(out, (N, Co, H, W), x0) else torch.empty((N, Co, H, W), dtype=x0.dtype, device=x0.device)
        ptrs = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        with torch.cuda.device(x0.device):
            if _wino43_ok(N, Ci, Co, H, W, len(parts), weight):      # wide layer: F(4x4, 3x3) (set_winograd_tile)
                U = _wino43_weights(weight, transposed)
                _native.check(L.tai_conv3x3_wino43_forward_parts(ptrs, len(parts), U.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Ci,
                                                                 Co, H, W, _ACT[act], torch.cuda.current_stream(x0.device).cuda_stream),
                              'tai_conv3x3_wino43_forward_parts')
                return y
            U = _wino_weights(weight, transposed)
            _native.check(L.tai_conv3x3_wino_forward_parts(ptrs, len(parts), U.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Ci,
                                                           Co, H, W, _ACT[act], torch.cuda.current_stream(x0.device).cuda_stream),
                          'tai_conv3x3_wino_forward_parts')
        return y
    fused = (x.is_cuda and x.dtype == torch.float32 and bias is not None
             and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad)))
    if not fused:
        if x.is_cuda and x.dtype == torch.float32 and bias is not None and weight.shape[2] == weight.shape[3] == 3:
            Co, Ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
            N, _, H, W = x.shape
            if _wino_ok(N, Ci, Co, H, W, 3, 3, padding) and _wino_ok(N, Co, Ci, H, W, 3, 3, padding):
                return _WinoConv3x3.apply(x, weight, bias, act, transposed)      # training: autograd through the HIP kernel
        if (x.is_cuda and x.dtype == torch.float32 and bias is not None and not transposed and weight.shape[2] == weight.shape[3]
                and weight.shape[2] in (5, 7)):
            Co, Ci, k = weight.shape[0], weight.shape[1], weight.shape[2]
            N, _, H, W = x.shape
            if _kxk_ok(N, Ci, Co, H, W, k, k, padding) and _kxk_ok(N, Co, Ci, H, W, k, k, padding):
                return _WinoConvKxK.apply(x, weight, bias, act)
        if x.is_cuda and x.dtype == torch.float32 and bias is not None and x.shape[3] % 4 == 0 and weight.shape[2] == weight.shape[3]:
            Co, Ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
            k = weight.shape[2]
            if Ci == 1 and not transposed and k in (3, 5) and padding == k // 2 and act in (None, 'relu') and Co >= 16:
                return _ThinInConv.apply(x, weight, bias, act)
            if Co == 1 and k == 3 and padding == 1 and Ci >= 16:
                return _ThinOutConv.apply(x, weight, bias, act, transposed)
        y = F.conv2d(x, _as_conv_weight(weight, transposed), bias, stride=1, padding=padding)
        return torch.relu(y) if act == 'relu' else (torch.tanh(y) if act == 'tanh' else y)
    kh, kw = weight.shape[2], weight.shape[3]
    Co, Ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
    N, _, H, W = x.shape
    L = _native.lib()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    thin_in = Ci == 1 and kh == kw and kh in (3, 5) and padding == kh // 2 and W % 4 == 0 and act in (None, 'relu') and Co >= 16
    thin_out = Co == 1 and kh == kw == 3 and padding == 1 and W % 4 == 0 and Ci >= 16
    if thin_in or thin_out:
        # one input or one output channel: no GEMM in it, a stream of the wide tensor (csrc/thin_conv.hip.inc)
        x = x.contiguous()
        w = _cached(weight, ('direct', transposed), lambda: _as_conv_weight(weight.detach(), transposed).contiguous())
        y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            if thin_in:
                _native.check(L.tai_conv_cin1_forward(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Co,
                                                      H, W, kh, _ACT[act], stream), 'tai_conv_cin1_forward')
            else:
                _native.check(L.tai_conv_cout1_3x3_forward(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), N,
                                                           Ci, H, W, _ACT[act], stream), 'tai_conv_cout1_3x3_forward')
        return y
    if not transposed and _kxk_ok(N, Ci, Co, H, W, kh, kw, padding):
        return _kxk_as_wino(x, weight, bias, act, False)
    if _wino_ok(N, Ci, Co, H, W, kh, kw, padding, min_ci=2):
        # Winograd F(2x2,3x3) on the fp32 MFMA pipe (csrc/wino_conv.hip.inc)
        x = x.contiguous()
        y = out if _usable_out(out, (N, Co, H, W), x) else torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            if _wino43_ok(N, Ci, Co, H, W, 1, weight):       # wide layer: F(4x4, 3x3) (set_winograd_tile)
                U = _wino43_weights(weight, transposed)
                _native.check(L.tai_conv3x3_wino43_forward(x.data_ptr(), U.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Ci, Co,
                                                           H, W, _ACT[act], stream), 'tai_conv3x3_wino43_forward')
                return y
            U = _wino_weights(weight, transposed)
            _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Ci, Co,
                                                     H, W, _ACT[act], stream), 'tai_conv3x3_wino_forward')
        return y
    w = _cached(weight, ('direct', transposed), lambda: _as_conv_weight(weight.detach(), transposed).contiguous()) \
        if transposed else weight
    y = F.conv2d(x, w, None, stride=1, padding=padding)
    if not y.is_contiguous():
        y = y.contiguous()
    N, C, H, W = y.shape
    with torch.cuda.device(y.device):
        _native.check(L.tai_bias_act_inplace(y.data_ptr(), bias.data_ptr(), N, C, H * W, _ACT[act], stream),
                      'tai_bias_act_inplace')
    return y
