"""Winograd F(2x2,3x3) MFMA convolution (csrc/wino_conv.hip.inc) against an fp64 convolution of the same operands.
Reference layers: nn.Conv2d(C, K, 3, padding=1) (+ReLU) of src/models/mcnet/mcnet.py:79-118,131-152,165-170,271 and
src/models/tai/tai.py:248-286; nn.ConvTranspose2d(C, K, 3, padding=1) of mcnet.py:198-224."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _wino(x, w, b, act):
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, H, W = x.shape
    K = w.shape[0]
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W,
                                             {None: 0, 'relu': 1, 'tanh': 2}[act], s), 'forward')
    return y


def _check(x, w, b, act, tol=4e-6):
    got = _wino(x, w, b, act)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = torch.relu(ref) if act == 'relu' else (torch.tanh(ref) if act == 'tanh' else ref)
    # fp32 Winograd: the error scales with the sum of |terms| of the dot product (the transforms add and subtract
    # neighbouring inputs and weights before multiplying)
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
    err = ((got.double() - ref).abs() / (1 + mag)).max().item()
    assert torch.isfinite(got).all()
    assert err <= tol, err


# (N, C, K, H, W): channel counts that are not multiples of the 8-channel chunk or the 64-channel block, tile counts
# that are not multiples of the 64-tile block, tiles that straddle images, 2x2 and 4x4 images, the bi-TAI shapes
SHAPES = [(1, 8, 8, 2, 2), (2, 8, 8, 8, 8), (3, 20, 51, 12, 20), (1, 65, 64, 16, 16), (5, 13, 70, 6, 10), (7, 9, 3, 4, 4),
          (2, 64, 64, 128, 128), (2, 51, 51, 128, 128), (4, 512, 128, 16, 16), (3, 128, 130, 32, 32)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
def test_wino_matches_fp64_conv(shape, act):
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    _check(x, w, b, act)


def test_wino_exact_on_small_integers():
    # integer-valued inputs and weights whose transforms stay exact in fp32: the result must be the exact convolution
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-4, 5, (2, 16, 12, 12), generator=g).float().cuda()
    w = (torch.randint(-2, 3, (32, 16, 3, 3), generator=g) * 4).float().cuda()     # multiples of 4: G g G^T exact
    b = torch.randint(-3, 4, (32,), generator=g).float().cuda()
    got = _wino(x, w, b, None)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert torch.equal(got.double(), ref)


def test_wino_zero_padding_not_replication():
    x = torch.ones(1, 8, 4, 4, device='cuda')
    w = torch.ones(8, 8, 3, 3, device='cuda')
    b = torch.zeros(8, device='cuda')
    got = _wino(x, w, b, None)
    assert got[0, 0, 0, 0].item() == 32.0 and got[0, 0, 0, 1].item() == 48.0 and got[0, 0, 1, 1].item() == 72.0


def test_conv_bias_act_routes_to_wino_and_tracks_weight_updates():
    from video_frame_inpainting_amd import conv_ops
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(16, 64, 64, 64, device='cuda')
    with torch.no_grad():
        y1 = conv_ops.conv_bias_act(x, conv.weight, conv.bias, 1, 'relu')
        assert ('wino', False, 0) in conv.weight._tai_derived                      # the MFMA kernel ran, not MIOpen
        assert (y1 - torch.relu(conv(x))).abs().max().item() <= 5e-5
        conv.weight.mul_(2.0)                                                     # in-place update -> U is rebuilt
        y2 = conv_ops.conv_bias_act(x, conv.weight, conv.bias, 1, 'relu')
        assert (y2 - torch.relu(conv(x))).abs().max().item() <= 1e-4


def test_transposed_conv_through_wino():
    from video_frame_inpainting_amd import conv_ops
    layer = torch.nn.ConvTranspose2d(128, 64, 3, padding=1).cuda()
    x = torch.randn(16, 128, 64, 64, device='cuda')
    with torch.no_grad():
        got = conv_ops.conv_bias_act(x, layer.weight, layer.bias, 1, 'relu', transposed=True)
        assert ('wino', True, 0) in layer.weight._tai_derived
        ref = torch.relu(layer(x))
    assert (got - ref).abs().max().item() <= 5e-5


def test_wino_rejects_bad_arguments():
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    x = torch.zeros(1, 8, 5, 4, device='cuda')
    U = torch.zeros(L.tai_conv3x3_wino_weight_floats(8, 8), device='cuda')
    b = torch.zeros(8, device='cuda')
    y = torch.zeros(1, 8, 5, 4, device='cuda')
    assert L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 8, 8, 5, 4, 0, None) != 0
    assert b'even' in L.tai_sepconv_last_error()
    assert L.tai_conv3x3_wino_forward(None, U.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 8, 8, 4, 4, 0, None) != 0
    assert L.tai_conv3x3_wino_weight_floats(64, 64) == 16 * 64 * 64
    assert L.tai_conv3x3_wino_weight_floats(51, 65) == 16 * 64 * 72


@pytest.mark.parametrize('nparts', [2, 4])
def test_wino_reads_channel_parts_without_a_cat(nparts):
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(5)
    parts = [torch.randn(32, 64, 32, 32, generator=g).cuda() for _ in range(nparts)]
    conv = torch.nn.Conv2d(64 * nparts, 128, 3, padding=1).cuda()
    with torch.no_grad():
        got = conv_ops.conv_bias_act(tuple(parts), conv.weight, conv.bias, 1, 'relu')
        assert ('wino', False, 0) in conv.weight._tai_derived
        whole = conv_ops.conv_bias_act(torch.cat(parts, dim=1), conv.weight, conv.bias, 1, 'relu')
    assert torch.equal(got, whole)                      # same kernel, same arithmetic, same order


def test_channel_parts_fall_back_to_cat_when_not_eligible():
    from video_frame_inpainting_amd import conv_ops
    a = torch.randn(2, 12, 8, 8, device='cuda')          # 12 channels per part: not a multiple of 8
    b = torch.randn(2, 12, 8, 8, device='cuda')
    conv = torch.nn.Conv2d(24, 16, 3, padding=1).cuda()
    with torch.no_grad():
        got = conv_ops.conv_bias_act((a, b), conv.weight, conv.bias, 1, None)
        ref = conv(torch.cat((a, b), dim=1))
    assert (got - ref).abs().max().item() <= 1e-5
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    assert L.tai_conv3x3_wino_forward_parts(None, 2, 0, 0, 0, 1, 16, 8, 4, 4, 0, None) != 0


@pytest.mark.parametrize('shape', [(16, 64, 64, 64, 64), (3, 24, 51, 12, 20)])
def test_wino_fused_maxpool(shape):
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(K, C, 3, 3, generator=g) * 0.1).cuda(); b = torch.randn(K, generator=g).cuda()
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, None), 'transform')
    y = torch.full((N, K, H, W), float('nan'), device='cuda'); yp = torch.full((N, K, H // 2, W // 2), float('nan'), device='cuda')
    _native.check(L.tai_conv3x3_wino_forward_maxpool(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), yp.data_ptr(), N, C, K, H, W, 1, None), 'fwd')
    assert torch.equal(y, _wino(x, w, b, 'relu'))
    assert torch.equal(yp, F.max_pool2d(y, 2))


def test_encoders_use_the_fused_pool_and_match_aten():
    from video_frame_inpainting_amd import conv_ops
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(16, 64, 64, 64, device='cuda')
    with torch.no_grad():
        y, yp = conv_ops.conv_bias_act_maxpool(x, conv.weight, conv.bias, 1, 'relu')
        assert torch.equal(yp, F.max_pool2d(y, 2)) and (y - torch.relu(conv(x))).abs().max().item() <= 5e-5
        c5 = torch.nn.Conv2d(64, 128, 5, padding=2).cuda()              # MIOpen path: pooled by ATen
        y, yp = conv_ops.conv_bias_act_maxpool(x, c5.weight, c5.bias, 2, 'relu')
        assert torch.equal(yp, F.max_pool2d(y, 2)) and (y - torch.relu(c5(x))).abs().max().item() <= 1e-4


@pytest.mark.parametrize('k,shape', [(5, (16, 64, 128, 64, 64)), (7, (16, 128, 256, 32, 32)), (5, (48, 16, 24, 20, 28))])
@pytest.mark.parametrize('pool', [False, True])
def test_5x5_and_7x7_through_the_winograd_kernel(k, shape, pool):
    """MotionEnc's 5x5 / 7x7 layers (mcnet.py:36-38, 45-47) as a 3x3 convolution over shifted copies of the input."""
    from video_frame_inpainting_amd import conv_ops
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(k * 100 + H)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    conv = torch.nn.Conv2d(C, K, k, padding=k // 2).cuda()
    with torch.no_grad():
        if pool:
            y, yp = conv_ops.conv_bias_act_maxpool(x, conv.weight, conv.bias, k // 2, 'relu')
            assert torch.equal(yp, F.max_pool2d(y, 2))
        else:
            y = conv_ops.conv_bias_act(x, conv.weight, conv.bias, k // 2, 'relu')
        assert ('wino_kxk', False, 0) in conv.weight._tai_derived                    # not the MIOpen path
        ref = torch.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=k // 2))
        mag = F.conv2d(x.double().abs(), conv.weight.double().abs(), conv.bias.double().abs(), padding=k // 2)
    err = ((y.double() - ref).abs() / (1 + mag)).max().item()
    assert err <= 4e-6, err


def test_shift_stack_layout():
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, H, W, k = 2, 3, 6, 8, 5
    x = torch.arange(N * C * H * W, dtype=torch.float32, device='cuda').view(N, C, H, W) + 1
    out = torch.full((N, 4 * C, H + 2, W + 4), float('nan'), device='cuda')
    _native.check(L.tai_conv_shift_stack(x.data_ptr(), out.data_ptr(), N, C, H, W, k, None), 'stack')
    for a in range(2):
        for b in range(2):
            oy, ox = 3 * a - 1, 3 * b - 1
            blk = out[:, (a * 2 + b) * C:(a * 2 + b + 1) * C]
            ref = torch.zeros_like(blk)
            for u in range(H + 2):
                for v in range(W + 4):
                    sy, sx = u - 1 + oy, v - 2 + ox
                    if 0 <= sy < H and 0 <= sx < W:
                        ref[:, :, u, v] = x[:, :, sy, sx]
            assert torch.equal(blk, ref)


@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
def test_training_form_gradients_match_autograd_of_conv2d(act, transposed):
    """Forward and input gradient on the Winograd kernel, weight / bias gradients from MIOpen, against fp64 autograd."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(17)
    N, C, K, H, W = 32, 64, 128, 32, 32
    x = torch.randn(N, C, H, W, generator=g).cuda().requires_grad_(True)
    wshape = (C, K, 3, 3) if transposed else (K, C, 3, 3)
    w = (torch.randn(*wshape, generator=g) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(K, generator=g).cuda().requires_grad_(True)
    go = torch.randn(N, K, H, W, generator=g).cuda()
    y = conv_ops.conv_bias_act(x, w, b, 1, act, transposed=transposed)
    assert type(y.grad_fn).__name__ == '_WinoConv3x3Backward'
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    weff = wd.transpose(0, 1).flip(2, 3) if transposed else wd
    yd = F.conv2d(xd, weff, bd, padding=1)
    yd = torch.relu(yd) if act == 'relu' else (torch.tanh(yd) if act == 'tanh' else yd)
    rx, rw, rb = torch.autograd.grad(yd, (xd, wd, bd), go.double())
    # y and gx are 576- / 1152-term fp32 sums; gw and gb sum over all 32,768 pixels (MIOpen, fp32), and a ReLU whose
    # pre-activation is within rounding of zero may open in one precision and not in the other
    for got, ref, tol in ((y, yd, 1e-5), (gx, rx, 2e-4), (gw, rw, 2e-4), (gb, rb, 2e-4)):
        err = (got.double() - ref).abs().max().item() / (1 + ref.abs().max().item())
        assert err <= tol, err


@pytest.mark.parametrize('shape', [(8, 64, 128, 32, 32), (3, 24, 256, 12, 20), (2, 130, 1024, 16, 16)])
def test_tall_workgroup_shape_gives_the_same_bits(shape):
    """K a multiple of 128: the 128-channel x 32-tile workgroup shape sums in the same order as the 64 x 64 one."""
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(K + C)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(K, generator=g).cuda()
    try:
        L.tai_conv3x3_wino_set_tall(0)
        ref = _wino(x, w, b, 'relu')
        L.tai_conv3x3_wino_set_tall(1)
        got = _wino(x, w, b, 'relu')
    finally:
        L.tai_conv3x3_wino_set_tall(1)
    assert torch.equal(got, ref)
    _check(x, w, b, 'relu')


# ---- round 2: displaced reads of a halo-carrying plane (5x5 / 7x7 without a stack of shifted copies), pooled output
# written into the next layer's plane, unpool + residual add as a second output ----------------------------------------
def test_motion_encoder_chain_matches_the_stage_by_stage_ops(monkeypatch):
    """MotionEnc (mcnet.py:14-60) through conv_ops.motion_enc_chain: every stage against ATen on the same operands in
    fp64, twice with different inputs (the cached planes' halos must still be zero the second time), and the displaced-
    read path against the shifted-copy path bit for bit (same channel order, same chunks)."""
    from video_frame_inpainting_amd import conv_ops
    from video_frame_inpainting_amd.mcnet import MotionEnc
    from video_frame_inpainting_amd import synthetic
    monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 1)      # (a speed heuristic: small problems stay on MIOpen)
    enc = synthetic.seeded_init(MotionEnc(16), 3).cuda()
    convs = [enc.dyn_conv1.convs()[0], enc.dyn_conv2.convs()[0], enc.dyn_conv3.convs()[0]]
    for seed in (1, 2):
        x = torch.randn(6, 1, 64, 96, generator=torch.Generator().manual_seed(seed)).cuda()
        with torch.no_grad():
            got = conv_ops.motion_enc_chain(x, *convs)
            assert got is not None
            p3, (c1, c2, c3) = got
            r1 = torch.relu(F.conv2d(x.double(), convs[0].weight.double(), convs[0].bias.double(), padding=2))
            r2 = torch.relu(F.conv2d(F.max_pool2d(r1, 2), convs[1].weight.double(), convs[1].bias.double(), padding=2))
            r3 = torch.relu(F.conv2d(F.max_pool2d(r2, 2), convs[2].weight.double(), convs[2].bias.double(), padding=3))
            for a, b in ((c1, r1), (c2, r2), (c3, r3), (p3, F.max_pool2d(r3, 2))):
                assert a.shape == b.shape
                assert float((a.double() - b).abs().max()) <= 2e-5 * float(b.abs().max())
            # the shifted-copy path of round 1 on the same pooled inputs: identical arithmetic
            y2, yp2 = conv_ops._kxk_as_wino(F.max_pool2d(c1, 2), convs[1].weight, convs[1].bias, 'relu', True)
            assert torch.equal(y2, c2)
            y3, yp3 = conv_ops._kxk_as_wino(yp2, convs[2].weight, convs[2].bias, 'relu', True)
            assert torch.equal(y3, c3) and torch.equal(yp3, p3)
            out = enc(x)                     # the module takes the chain
            assert torch.equal(out[0], p3)


def test_motion_encoder_chain_declines_what_it_cannot_run(monkeypatch):
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 1)
    from video_frame_inpainting_amd.mcnet import MotionEnc
    enc = MotionEnc(4).cuda()                 # 4 channels: not a multiple of the 8-channel chunk
    convs = [enc.dyn_conv1.convs()[0], enc.dyn_conv2.convs()[0], enc.dyn_conv3.convs()[0]]
    x = torch.randn(2, 1, 32, 32).cuda()
    with torch.no_grad():
        assert conv_ops.motion_enc_chain(x, *convs) is None
        p3, res = enc(x)                      # falls back to the stage-by-stage path
    assert p3.shape == (2, 16, 4, 4)
    enc16 = MotionEnc(16).cuda()
    c16 = [enc16.dyn_conv1.convs()[0], enc16.dyn_conv2.convs()[0], enc16.dyn_conv3.convs()[0]]
    assert conv_ops.motion_enc_chain(torch.randn(6, 1, 64, 96).cuda().requires_grad_(), *c16) is None     # autograd: stock ops


@pytest.mark.parametrize('shape', [(3, 64, 64, 32, 32), (2, 16, 128, 64, 64), (5, 24, 51, 12, 20)])
@pytest.mark.parametrize('nparts', [1, 2])
def test_unpool_add_second_output(shape, nparts, monkeypatch):
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 1)
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(C + K)
    parts = [torch.randn(N, C // nparts, H, W, generator=g).cuda() for _ in range(nparts)]
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    below = torch.randn(N, K, H // 2, W // 2, generator=g).cuda()
    with torch.no_grad():
        y, y2 = conv_ops.conv_bias_unpool_add(tuple(parts) if nparts > 1 else parts[0], w, b, 1, below)
        ref = conv_ops.conv_bias_act(torch.cat(parts, 1), w, b, 1, None)
    want = ref.clone()
    want[:, :, 0::2, 0::2] += below
    assert torch.equal(y, ref)                # same kernel, same arithmetic
    assert torch.equal(y2, want)
    with torch.no_grad():                     # only the sum wanted: it is written in place of the plain output
        none, only = conv_ops.conv_bias_unpool_add(tuple(parts) if nparts > 1 else parts[0], w, b, 1, below, keep_plain=False)
        assert none is None and torch.equal(only, want)
        buf = torch.full((2 * N, K, H, W), float('nan'), device='cuda')            # result written into a batch slice
        got = conv_ops.conv_bias_act(tuple(parts) if nparts > 1 else parts[0], w, b, 1, None, out=buf[N:])
        assert got.data_ptr() == buf[N:].data_ptr() and torch.equal(buf[N:], ref) and bool(torch.isnan(buf[:N]).all())


def test_general_entry_point_rejects_bad_arguments():
    import ctypes
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    t = torch.zeros(1 << 16, device='cuda')
    p = t.data_ptr()
    xs = (ctypes.c_void_p * 1)(p)
    s = torch.cuda.current_stream().cuda_stream
    ok = dict(nparts=1, shift=5, N=1, C=4 * 8, K=8, H=8, W=8, in_h=13, in_w=16, in_oy=1, in_ox=2)

    def call(**kw):
        a = dict(ok, **kw)
        return L.tai_conv3x3_wino_forward_ex(xs, a['nparts'], a['shift'], p, p, p, None, 0, 0, 0, 0, None, None, a['N'], a['C'], a['K'],
                                             a['H'], a['W'], a['in_h'], a['in_w'], a['in_oy'], a['in_ox'], a.get('act', 1), s)
    assert call() == 0
    assert call(act=0) == -1                                             # displaced reads are built with ReLU only
    assert call(shift=3) == -1 and call(shift=10) == -1 and call(C=4 * 6) == -1     # 4 <= shift_k <= 9; C / S^2 a multiple of 8
    assert call(in_h=12) == -1 and call(in_w=14) == -1 and call(in_ox=0) == -1      # the plane must hold the halo
    assert L.tai_conv3x3_wino_forward_ex(xs, 1, 0, p, p, p, None, 0, 0, 0, 0, None, p, 1, 8, 8, 8, 8, 8, 8, 0, 0, 0, s) == -1   # y2 without addx
    torch.cuda.synchronize()


# ---- Winograd-domain weight gradient (csrc/wino_wrw.hip.inc): dL/dw of nn.Conv2d(C, K, 3, padding=1) under loss.backward()
# (N, C, K, H, W): channel counts that are not multiples of 8 / 16 / 64, several 64-blocks, one chunk per split, many
# chunks per split, tile rows of exactly one chunk (W = 16), the bi-TAI shapes
WRW_SHAPES = [(1, 8, 8, 2, 16), (2, 16, 16, 8, 16), (3, 20, 51, 12, 32), (1, 65, 64, 16, 16), (5, 13, 70, 6, 48), (7, 9, 3, 4, 16),
              (2, 64, 64, 128, 128), (4, 512, 128, 16, 16), (3, 128, 130, 32, 32), (32, 64, 64, 64, 64),
              (6, 24, 40, 8, 8), (5, 16, 9, 4, 4), (3, 8, 8, 2, 2), (2, 40, 24, 6, 12)]      # rows of fewer than 16 pixels: widened with zeros


def _weight_grad_fp64(x, go):
    xd = x.double()
    wd = torch.zeros(go.shape[1], x.shape[1], 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    return torch.autograd.grad(F.conv2d(xd, wd, None, padding=1), wd, go.double())[0]


@pytest.fixture
def wrw_tile_2():
    """the F(2x2, 3x3)-domain weight-gradient kernel for the tests of ITS properties (exact on integers, load schemes, input windows)"""
    from video_frame_inpainting_amd import conv_ops
    prev = conv_ops.set_weight_gradient_tile(2)
    yield
    conv_ops.set_weight_gradient_tile(prev)


@pytest.mark.parametrize('tile', [4, 2])
@pytest.mark.parametrize('shape', WRW_SHAPES)
def test_wino_weight_gradient_matches_fp64(shape, tile):
    """tile 4 (default): the F(4x4, 3x3)-domain kernel where H % 4 == 0 (conv3x3_wrw_gen; other shapes fall to the F(2x2, 3x3) kernel, and
    rows shorter than 16 pixels are widened first), tile 2: the F(2x2, 3x3) kernel everywhere."""
    from video_frame_inpainting_amd import conv_ops
    N, C, K, H, W = shape
    prev = conv_ops.set_weight_gradient_tile(tile)
    try:
        _weight_gradient_matches_fp64(N, C, K, H, W)
    finally:
        assert conv_ops.set_weight_gradient_tile(prev) == tile


def _weight_gradient_matches_fp64(N, C, K, H, W):
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    go = torch.randn(N, K, H, W, generator=g).cuda()
    got = conv_ops.wino_weight_grad(x, go)
    assert got is not None and got.shape == (K, C, 3, 3) and torch.isfinite(got).all()
    ref = _weight_grad_fp64(x, go)
    # sums of N*H*W products of unit-variance terms: the fp32 error scales with sqrt(count) times the term size
    scale = (N * H * W) ** 0.5
    err = (got.double() - ref).abs().max().item() / scale
    assert err <= 2e-5, err
    again, gb = conv_ops.wino_weight_grad(x, go, with_bias=True)
    assert torch.equal(got, again)                      # fixed reduction order: reproducible
    # the bias gradient summed by the same kernel
    rb = go.double().sum((0, 2, 3))
    assert gb.shape == (K,) and float((gb.double() - rb).abs().max()) / scale <= 2e-5
    assert torch.equal(gb, conv_ops.wino_weight_grad(x, go, with_bias=True)[1])


def test_wino_weight_gradient_paired_and_plain_chunks_agree(wrw_tile_2):
    """W % 32 == 0 takes the paired-chunk load scheme (whole 128-byte lines); the 8-tile scheme on the same operands sums the
    same products in another order."""
    from video_frame_inpainting_amd import _native, conv_ops
    L = _native.lib()
    g = torch.Generator().manual_seed(31)
    for (N, C, K, H, W) in ((3, 24, 40, 16, 32), (2, 64, 64, 64, 64), (5, 13, 70, 6, 96)):
        x = torch.randn(N, C, H, W, generator=g).cuda()
        go = torch.randn(N, K, H, W, generator=g).cuda()
        paired, pb = conv_ops.wino_weight_grad(x, go, with_bias=True)
        prev = L.tai_conv3x3_wino_wrw_set_paired(0)
        try:
            plain, qb = conv_ops.wino_weight_grad(x, go, with_bias=True)
        finally:
            L.tai_conv3x3_wino_wrw_set_paired(prev)
        scale = (N * H * W) ** 0.5
        assert float((paired - plain).abs().max()) / scale <= 2e-5
        assert float((pb - qb).abs().max()) / scale <= 2e-5


def test_wino_weight_gradient_exact_on_small_integers(wrw_tile_2):
    """Integer-valued operands keep every Winograd intermediate exact in fp32 (the transforms only add, the 1/2 and 1/4
    factors of G^T . G are exact), so the result must equal the integer sums bit for bit."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(5)
    N, C, K, H, W = 3, 24, 40, 16, 32
    x = torch.randint(-3, 4, (N, C, H, W), generator=g).float().cuda()
    go = torch.randint(-2, 3, (N, K, H, W), generator=g).float().cuda()
    got = conv_ops.wino_weight_grad(x, go)
    ref = _weight_grad_fp64(x, go)
    assert torch.equal(got.double(), ref)


@pytest.mark.parametrize('tile', [4, 2])
def test_wino_weight_gradient_sees_the_zero_padding_and_every_tap(tile):
    """One non-zero output-gradient pixel in a corner, on an edge and in the interior: dw[k, c, a, b] = x[c, y + a - 1, x + b - 1]
    with zeros outside the image -- the F(4x4, 3x3) kernel gets there without a halo (rows -1 / H through a zero-size buffer descriptor,
    the columns left / right of a tile row through the coefficient of the edge value; two images and W = 32, so that chunks start, end
    and lie inside tile rows) and is exact to rounding only (its constants are multiples of 1/64 and 1/81)."""
    from video_frame_inpainting_amd import conv_ops
    N, C, K, H, W = 2, 8, 8, 8, 32
    x = torch.arange(N * C * H * W, dtype=torch.float32).view(N, C, H, W).cuda() % 17 - 8
    prev = conv_ops.set_weight_gradient_tile(tile)
    try:
        for n in (0, 1):
            for (py, px) in ((0, 0), (H - 1, W - 1), (3, 7), (0, 9), (5, 0), (4, 15), (3, 16), (7, 31), (0, 31), (7, 0), (4, 4)):
                go = torch.zeros(N, K, H, W, device='cuda')
                go[n, 2, py, px] = 1.0
                got = conv_ops.wino_weight_grad(x, go)
                xp = F.pad(x, (1, 1, 1, 1))
                want = torch.zeros(K, C, 3, 3, device='cuda')
                want[2] = xp[n, :, py:py + 3, px:px + 3]
                if tile == 2:
                    assert torch.equal(got, want), (n, py, px)
                else:
                    assert float((got - want).abs().max()) <= 2e-5, (n, py, px, float((got - want).abs().max()))
    finally:
        conv_ops.set_weight_gradient_tile(prev)


def test_wino_weight_gradient_declines_unsupported_shapes():
    from video_frame_inpainting_amd import _native, conv_ops
    x = torch.randn(1, 8, 6, 24, device='cuda')                   # W above 16 and not a multiple of 16
    assert conv_ops.wino_weight_grad(x, torch.randn(1, 8, 6, 24, device='cuda')) is None
    x = torch.randn(1, 8, 5, 16, device='cuda')                   # odd H
    assert conv_ops.wino_weight_grad(x, torch.randn(1, 8, 5, 16, device='cuda')) is None
    L = _native.lib()
    assert L.tai_conv3x3_wino_wrw_workspace_floats(64, 512, 64, 128, 128) == -1       # 2 GiB input
    assert L.tai_conv3x3_wino_wrw(None, None, None, None, None, 1, 8, 8, 4, 16, None) != 0


@pytest.mark.parametrize('k,shape', [(5, (16, 64, 128, 64, 64)), (7, (16, 128, 256, 32, 32))])
@pytest.mark.parametrize('act', [None, 'relu'])
@pytest.mark.parametrize('tile', [2, 4])
def test_5x5_and_7x7_training_form_gradients_match_autograd_of_conv2d(k, shape, act, tile, monkeypatch):
    """MotionEnc's 5x5 / 7x7 layers under autograd: forward and input gradient through the Winograd kernel -- F(2x2, 3x3) over shifted
    copies, or (tile 4, the default) F(4x4, 3x3) blocks displaced over a halo plane -- the input gradient with the transposed, flipped
    filter; weight / bias gradients from the Winograd-domain weight-gradient kernel over the stack of shifted copies."""
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO43_UNDER_AUTOGRAD', tile == 4)
    monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)          # (these 16-image shapes are below the dispatch threshold)
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(k * 10 + H)
    x = torch.randn(N, C, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(K, C, k, k, generator=g) * 0.03).cuda().requires_grad_(True)
    b = torch.randn(K, generator=g).cuda().requires_grad_(True)
    go = torch.randn(N, K, H, W, generator=g).cuda()
    y = conv_ops.conv_bias_act(x, w, b, k // 2, act)
    assert type(y.grad_fn).__name__ == '_WinoConvKxKBackward'
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    assert (('wino43_kxk', True) if tile == 4 else ('wino_kxk', True, 0)) in w._tai_derived
    assert (('wino43_kxk', False) if tile == 4 else ('wino_kxk', False, 0)) in w._tai_derived
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = F.conv2d(xd, wd, bd, padding=k // 2)
    # the ReLU's mask is taken from the fp32 output: a pre-activation within rounding of zero (there are a few among the
    # 4M outputs of 6,272-term sums) may open in one precision and not in the other, and that is not what is tested here
    yd = yd * (y.detach() > 0) if act == 'relu' else yd
    rx, rw, rb = torch.autograd.grad(yd, (xd, wd, bd), go.double())
    for got, ref, tol in ((y, yd, 1e-5 if tile == 2 else 6e-5), (gx, rx, 2e-4), (gw, rw, 2e-4), (gb, rb, 2e-4)):
        err = (got.double() - ref).abs().max().item() / (1 + ref.abs().max().item())
        assert err <= tol, err


@pytest.mark.parametrize('tile', [4, 2])
def test_wino_weight_gradient_reads_a_haloed_plane_like_the_padded_tensor(tile):
    """tai_conv3x3_wino_wrw_window on a zero-framed copy of x (origin (1, 2) / (3, 4)) sums exactly the products of the
    plain entry on x: the frame's zeros stand where the plain entry pads.  (F(2x2, 3x3) kernel: the same bits; the F(4x4, 3x3) kernel
    multiplies the halo's zeros where the plain entry skips the loads: equal to rounding.)"""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(41)
    prev = conv_ops.set_weight_gradient_tile(tile)
    try:
        _haloed_plane_checks(conv_ops, g, tile)
    finally:
        conv_ops.set_weight_gradient_tile(prev)


def _haloed_plane_checks(conv_ops, g, tile):
    for (N, C, K, H, W), (oy, ox) in (((3, 24, 40, 16, 32), (1, 2)), ((2, 16, 16, 8, 16), (3, 4)), ((2, 64, 72, 32, 64), (1, 2))):
        x = torch.randn(N, C, H, W, generator=g).cuda()
        go = torch.randn(N, K, H, W, generator=g).cuda()
        plane = torch.zeros(N, C, H + 2 * oy, W + 2 * ox, device='cuda')
        plane[:, :, oy:oy + H, ox:ox + W] = x
        a, ab = conv_ops.wino_weight_grad(x, go, with_bias=True)
        b, bb = conv_ops.wino_weight_grad(plane, go, with_bias=True, window=(oy, ox))
        if tile == 2:
            assert torch.equal(a, b) and torch.equal(ab, bb)
        else:
            assert torch.equal(ab, bb) and float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
            ref = _weight_grad_fp64(x, go)
            assert float((b.double() - ref).abs().max()) / (N * H * W) ** 0.5 <= 2e-5
    # a non-zero frame is read, not padded over
    plane = torch.randn(2, 16, 8 + 2, 16 + 4, generator=g).cuda()
    go = torch.randn(2, 16, 8, 16, generator=g).cuda()
    got = conv_ops.wino_weight_grad(plane, go, window=(1, 2))
    wd = torch.zeros(16, 16, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(plane.double()[:, :, :, 1:-1], wd, None), wd, go.double())[0]     # valid convolution over the framed plane
    assert float((got.double() - ref).abs().max()) <= 2e-4


@pytest.mark.parametrize('nparts,cp,co,act,transposed', [(2, 64, 64, 'relu', False), (4, 32, 128, None, False), (2, 64, 64, None, True),
                                                          (3, 24, 64, 'tanh', False)])
def test_training_form_on_channel_parts_matches_the_concatenating_path_and_fp64(nparts, cp, co, act, transposed, monkeypatch):
    """conv_ops._WinoConv3x3Parts (training: a tuple of channel parts read where they lie, one contiguous input gradient per part,
    weight gradient per part) against the path that concatenates first (_WinoConv3x3) and against autograd of an fp64 conv2d.
    Reference: conv(cat(...)) of Residual / CombLayers / ConvLstmCell, src/models/mcnet/mcnet.py:131-176,271-293."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(nparts * 100 + cp)
    N, H, W = 8, 64, 64
    ci = nparts * cp
    parts0 = [torch.randn(N, cp, H, W, generator=g).cuda() for _ in range(nparts)]
    wshape = (ci, co, 3, 3) if transposed else (co, ci, 3, 3)
    w0 = (torch.randn(*wshape, generator=g) * (2.0 / (9 * ci)) ** 0.5).cuda()
    b0 = torch.randn(co, generator=g).cuda()
    go = torch.randn(N, co, H, W, generator=g).cuda()

    def run(parts_path):
        monkeypatch.setattr(conv_ops, 'PARTS_UNDER_AUTOGRAD', parts_path)
        parts = [p.clone().requires_grad_() for p in parts0]
        w, b = w0.clone().requires_grad_(), b0.clone().requires_grad_()
        y = conv_ops.conv_bias_act(tuple(parts), w, b, 1, act, transposed=transposed)
        y.backward(go)
        return y.detach(), [p.grad for p in parts], w.grad, b.grad

    y1, gp1, gw1, gb1 = run(True)
    y0, gp0, gw0, gb0 = run(False)
    assert torch.equal(y1, y0)                                         # the parts kernel: the same bits as the concatenation
    for a, c in zip(gp1, gp0):
        assert a.is_contiguous() and torch.equal(a, c)                 # K-slices of the same input-gradient convolution
    assert float((gw1 - gw0).abs().max()) <= 2e-5 * float(gw0.abs().max())     # per-part partial sums are grouped differently
    assert float((gb1 - gb0).abs().max()) <= 2e-5 * float(gb0.abs().max())
    # fp64 autograd of the reference form
    parts = [p.double().requires_grad_() for p in parts0]
    w, b = w0.double().requires_grad_(), b0.double().requires_grad_()
    wc = w.transpose(0, 1).flip(2, 3) if transposed else w
    y = F.conv2d(torch.cat(parts, 1), wc, b, padding=1)
    y = torch.relu(y) if act == 'relu' else (torch.tanh(y) if act == 'tanh' else y)
    y.backward(go.double())
    assert float((y1.double() - y.detach()).abs().max()) <= 2e-5 * float(y.detach().abs().max())
    for a, c in zip(gp1, parts):
        assert float((a.double() - c.grad).abs().max()) <= 2e-5 * float(c.grad.abs().max())
    assert float((gw1.double() - w.grad).abs().max()) <= 5e-5 * float(w.grad.abs().max())
    assert float((gb1.double() - b.grad).abs().max()) <= 5e-5 * float(b.grad.abs().max())
