"""Split-bf16 arithmetic of the Winograd 3x3 convolution (csrc/wino_split.hip.inc; opt-in, tai_conv3x3_wino_set_arithmetic(1))
against an fp64 convolution of the same operands and against the fp32 MFMA kernel: three bf16 terms per operand, six bf16
products per product, fp32 accumulation -- the bound is the fp32 kernel's own (tests/test_gpu_wino_conv.py).
Reference layers: nn.Conv2d(C, K, 3, padding=1) (+ReLU) of src/models/mcnet/mcnet.py:79-118,131-152,165-170,271 and
src/models/tai/tai.py:248-286."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

_ACT = {None: 0, 'relu': 1, 'tanh': 2}


@pytest.fixture
def split_mode():
    from video_frame_inpainting_amd import conv_ops
    prev = conv_ops.set_winograd_arithmetic('bf16x3')
    yield
    conv_ops.set_winograd_arithmetic(prev)


def _weights(w):
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    K, C = w.shape[0], w.shape[1]
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, torch.cuda.current_stream().cuda_stream), 'transform')
    return U


def _wino(x, w, b, act, U=None):
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, H, W = x.shape
    K = w.shape[0]
    U = _weights(w) if U is None else U
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, _ACT[act],
                                             torch.cuda.current_stream().cuda_stream), 'forward')
    return y


def _err(got, x, w, b, act):
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = torch.relu(ref) if act == 'relu' else (torch.tanh(ref) if act == 'tanh' else ref)
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
    assert torch.isfinite(got).all()
    return ((got.double() - ref).abs() / (1 + mag)).max().item()


# (N, C, K, H, W): as tests/test_gpu_wino_conv.py -- ragged channel and tile counts, tiles that straddle images, tile rows of
# 2, 4, 8, 16, 32 and 64 tiles (the split kernel's shapes) and of 3, 5, 6 and 10 (which stay on the fp32 kernel, same buffer)
SHAPES = [(1, 8, 8, 4, 4), (2, 8, 8, 8, 8), (3, 20, 51, 12, 16), (1, 65, 64, 16, 16), (5, 13, 70, 6, 8), (7, 9, 3, 4, 4),
          (2, 64, 64, 128, 128), (2, 51, 51, 128, 128), (4, 512, 128, 16, 16), (3, 128, 130, 32, 32), (2, 16, 64, 8, 64),
          (3, 20, 51, 12, 20), (5, 13, 70, 6, 10), (2, 24, 8, 6, 6), (1, 8, 8, 2, 2)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
def test_split_matches_fp64_conv(shape, act, split_mode):
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    assert _err(_wino(x, w, b, act), x, w, b, act) <= 4e-6          # the fp32 kernel's bound


def test_split_error_is_at_or_below_the_fp32_kernels(split_mode):
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(11)
    for (N, C, K, H, W) in [(4, 64, 64, 64, 64), (2, 256, 128, 32, 32), (2, 512, 512, 16, 16)]:
        x = torch.randn(N, C, H, W, generator=g).cuda()
        w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
        b = torch.randn(K, generator=g).cuda()
        e_split = _err(_wino(x, w, b, None), x, w, b, None)
        conv_ops.set_winograd_arithmetic('fp32')
        try:
            e_fp32 = _err(_wino(x, w, b, None), x, w, b, None)
        finally:
            conv_ops.set_winograd_arithmetic('bf16x3')
        assert e_split <= 1.25 * e_fp32 + 1e-8, (e_split, e_fp32)


def test_split_exact_on_small_integers(split_mode):
    # integer inputs and weights whose transforms are exact in fp32 and whose three bf16 terms hold them exactly
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-4, 5, (2, 16, 12, 16), generator=g).float().cuda()
    w = (torch.randint(-2, 3, (32, 16, 3, 3), generator=g) * 4).float().cuda()
    b = torch.randint(-3, 4, (32,), generator=g).float().cuda()
    got = _wino(x, w, b, None)
    assert torch.equal(got.double(), F.conv2d(x.double(), w.double(), b.double(), padding=1))


def test_split_zero_padding_and_wide_rows(split_mode):
    # ones everywhere: interior 9 C, edges 6 C, corners 4 C -- on tile rows of 64 tiles (the 16-lane rows' own outer loads)
    x = torch.ones(1, 8, 4, 128, device='cuda')
    w = torch.ones(8, 8, 3, 3, device='cuda')
    b = torch.zeros(8, device='cuda')
    got = _wino(x, w, b, None)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert torch.equal(got.double(), ref)
    # a ramp along x: every column distinct, so a wrong neighbour shows
    x = torch.arange(128, device='cuda', dtype=torch.float32).view(1, 1, 1, 128).expand(2, 8, 6, 128).contiguous()
    got = _wino(x, w, b, None)
    assert torch.equal(got.double(), F.conv2d(x.double(), w.double(), b.double(), padding=1))


def test_buffer_decides_the_kernel_not_the_mode():
    """A buffer made in fp32 mode is read by the fp32 kernel whatever the mode is at launch time, and vice versa."""
    from video_frame_inpainting_amd import conv_ops, _native
    L = _native.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 32, 32, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(64, generator=g).cuda()
    assert conv_ops.get_winograd_arithmetic() == 'fp32'
    U32 = _weights(w)
    y32 = _wino(x, w, b, 'relu', U32)
    prev = conv_ops.set_winograd_arithmetic('bf16x3')
    try:
        assert L.tai_conv3x3_wino_weight_floats(64, 64) == 40 * 64 * 64
        Us = _weights(w)
        assert torch.equal(_wino(x, w, b, 'relu', U32), y32)             # fp32 buffer, split mode: fp32 kernel, bit for bit
        ys = _wino(x, w, b, 'relu', Us)
    finally:
        conv_ops.set_winograd_arithmetic(prev)
    assert torch.equal(_wino(x, w, b, 'relu', Us), ys)                   # split buffer, fp32 mode: split kernel
    assert not torch.equal(ys, y32) and (ys - y32).abs().max().item() < 1e-4
    assert L.tai_conv3x3_wino_set_arithmetic(7) < 0


@pytest.mark.parametrize('nparts', [2, 4])
def test_split_reads_channel_parts_without_a_cat(nparts, split_mode):
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(nparts)
    parts = [torch.randn(32, 64, 32, 32, generator=g).cuda() for _ in range(nparts)]
    conv = torch.nn.Conv2d(64 * nparts, 128, 3, padding=1).cuda()
    with torch.no_grad():
        got = conv_ops.conv_bias_act(tuple(parts), conv.weight, conv.bias, 1, 'relu')
        whole = conv_ops.conv_bias_act(torch.cat(parts, dim=1), conv.weight, conv.bias, 1, 'relu')
        x = torch.cat(parts, dim=1)
    assert torch.equal(got, whole)
    assert _err(got, x, conv.weight.detach(), conv.bias.detach(), 'relu') <= 4e-6


@pytest.mark.parametrize('shape', [(16, 64, 64, 64, 64), (16, 128, 128, 32, 32), (4, 64, 51, 128, 128)])
def test_split_pool_and_unpool_add_outputs(shape, split_mode):
    from video_frame_inpainting_amd import conv_ops
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    below = torch.randn(N, K, H // 2, W // 2, generator=g).cuda()
    with torch.no_grad():
        y, yp = conv_ops.conv_bias_act_maxpool(x, w, b, 1, 'relu')
        assert _err(y, x, w, b, 'relu') <= 4e-6
        assert torch.equal(yp, F.max_pool2d(y, 2))
        plain, summed = conv_ops.conv_bias_unpool_add(x, w, b, 1, below)
        assert _err(plain, x, w, b, None) <= 4e-6
        up = torch.zeros_like(plain)
        up[:, :, ::2, ::2] = below
        assert torch.equal(summed, plain + up)
        none, only = conv_ops.conv_bias_unpool_add(x, w, b, 1, below, keep_plain=False)
        assert none is None and torch.equal(only, summed)


def test_split_full_width_forward_matches_cpu_oracle(split_mode):
    """The whole bi-TAI forward (TAI_gray, full width, 128x128, K = F = T = 5) in split arithmetic against the CPU oracle, with the
    bound of the fp32 path's own test (tests/test_gpu_model.py: 1e-4 of each output's magnitude), and against the fp32-MFMA
    forward of the same weights."""
    import numpy as np
    import video_frame_inpainting_amd as vfi
    from video_frame_inpainting_amd import conv_ops, metrics, synthetic
    from oracle import tai_oracle
    m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    clips = synthetic.make_clips(2, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, 5, 5))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, 1, 5, 51, 5, P, Fo)
        m.to('cuda:0').eval()
        o_split = m(5, P.cuda(), Fo.cuda())
        conv_ops.set_winograd_arithmetic('fp32')
        try:
            o_fp32 = m(5, P.cuda(), Fo.cuda())
        finally:
            conv_ops.set_winograd_arithmetic('bf16x3')
    for k in ('pred', 'pred_forward', 'pred_backward', 'interp_net_outputs_1', 'interp_net_outputs_2'):
        scale = float(ref[k].abs().max())
        e_split = float((o_split[k].cpu() - ref[k]).abs().max()), float((o_fp32[k].cpu() - ref[k]).abs().max())
        assert e_split[0] <= 1e-4 * scale, (k, e_split, scale)
        assert not torch.equal(o_split[k], o_fp32[k])            # the split kernels did run
    p_s, s_s, _ = metrics.compute_errors(o_split['pred'].cpu().numpy(), GT.numpy())
    p_r, s_r, _ = metrics.compute_errors(ref['pred'].numpy(), GT.numpy())
    assert np.abs(np.asarray(p_s) - np.asarray(p_r)).max() <= 0.01 and np.abs(np.asarray(s_s) - np.asarray(s_r)).max() <= 1e-4


@pytest.mark.parametrize('shape', [(64, 64, 64, 128, 128), (64, 256, 128, 32, 32), (160, 512, 512, 4, 4), (7, 24, 70, 12, 16)])
def test_split_kernel_is_bit_reproducible(shape, split_mode):
    """The split kernel synchronises its eight waves with two barriers per chunk and hands patches from producers to consumers
    through ONE LDS stage per group; a stale or early read would show as a launch that differs from the first.  300 launches per
    shape, every one compared bit for bit (the kernel has no atomics and a fixed summation order)."""
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    U = _weights(w)
    first = _wino(x, w, b, 'relu', U)
    y = torch.empty_like(first)
    s = torch.cuda.current_stream().cuda_stream
    bad = 0
    for i in range(300):
        y.fill_(float('nan'))
        _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s), 'forward')
        bad += int(not torch.equal(y, first))
    assert bad == 0, bad


def test_arithmetics_are_cached_side_by_side_and_registry_entries_die_with_their_buffers():
    """a switch of the arithmetic used to drop the other arithmetic's transformed weights (a captured graph would then have read
    freed memory), and the library's record of split-layout buffers grew without bound."""
    import gc
    from video_frame_inpainting_amd import _native, conv_ops
    L = _native.lib()
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(8, 64, 64, 64, device='cuda')
    with torch.no_grad():
        conv_ops.conv_bias_act(x, conv.weight, conv.bias, 1, 'relu')
        u_fp32 = conv.weight._tai_derived[('wino', False, 0)][1]
        prev = conv_ops.set_winograd_arithmetic('bf16x3')
        try:
            conv_ops.conv_bias_act(x, conv.weight, conv.bias, 1, 'relu')
            u_split = conv.weight._tai_derived[('wino', False, 1)][1]
            assert conv.weight._tai_derived[('wino', False, 0)][1] is u_fp32          # still alive next to the split image
            assert u_split.numel() > u_fp32.numel()
        finally:
            conv_ops.set_winograd_arithmetic(prev)
    ptr = u_split.data_ptr()
    del u_split
    conv.weight._tai_derived.clear()
    gc.collect()
    assert L.tai_conv3x3_wino_forget_weights(ptr) == 0          # the finalizer already dropped the record
