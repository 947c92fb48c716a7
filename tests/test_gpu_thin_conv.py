"""Thin-layer direct convolutions (csrc/thin_conv.hip.inc) against an fp64 convolution of the same operands and against
ATen/MIOpen's fp32 one.  Reference layers: src/models/mcnet/mcnet.py:28-31 (1 -> gf, 5x5), :79-81 (1 -> gf, 3x3), :223-224
(gf -> 1 transposed 3x3 + tanh)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from video_frame_inpainting_amd import _native
    return _native


def _close(got, x, w, b, pad, act, tol=2e-6):
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
    ref = torch.relu(ref) if act == 'relu' else (torch.tanh(ref) if act == 'tanh' else ref)
    # tolerance scales with the sum of |terms| of the dot product, the quantity fp32 rounding is relative to
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=pad)
    err = ((got.double() - ref).abs() / (1 + mag)).max().item()
    assert err <= tol, err


@pytest.mark.parametrize('k', [3, 5])
@pytest.mark.parametrize('act', [None, 'relu'])
@pytest.mark.parametrize('shape', [(2, 64, 32, 32), (3, 16, 20, 36), (1, 64, 128, 128)])
def test_cin1_matches_conv2d(k, act, shape):
    from video_frame_inpainting_amd.conv_ops import conv_bias_act
    N, Co, H, W = shape
    g = torch.Generator().manual_seed(7 * k + H)
    x = torch.randn(N, 1, H, W, generator=g).cuda()
    w = (torch.randn(Co, 1, k, k, generator=g) * 0.3).cuda()
    b = torch.randn(Co, generator=g).cuda()
    with torch.no_grad():
        got = conv_bias_act(x, w, b, k // 2, act)
    assert got.shape == (N, Co, H, W)
    _close(got, x, w, b, k // 2, act)
    aten = F.conv2d(x, w, b, padding=k // 2)
    aten = torch.relu(aten) if act else aten
    assert (got - aten).abs().max().item() <= 2e-5


@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
@pytest.mark.parametrize('shape', [(2, 64, 32, 32), (3, 16, 20, 36), (1, 64, 128, 128), (2, 16, 9, 96), (1, 17, 5, 4)])
def test_cout1_matches_conv2d(act, shape):
    from video_frame_inpainting_amd.conv_ops import conv_bias_act
    N, Ci, H, W = shape
    g = torch.Generator().manual_seed(11 + H)
    x = torch.randn(N, Ci, H, W, generator=g).cuda()
    w = (torch.randn(1, Ci, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(1, generator=g).cuda()
    with torch.no_grad():
        got = conv_bias_act(x, w, b, 1, act)
    assert got.shape == (N, 1, H, W)
    _close(got, x, w, b, 1, act)


def test_cout1_is_the_transposed_conv_of_the_decoder():
    from video_frame_inpainting_amd.mcnet import _convt3x3_as_conv
    layer = torch.nn.ConvTranspose2d(64, 1, 3, padding=1).cuda()
    x = torch.randn(2, 64, 32, 32, device='cuda')
    with torch.no_grad():
        got = _convt3x3_as_conv(x, layer, 'tanh')
        ref = torch.tanh(layer(x))
    assert (got - ref).abs().max().item() <= 2e-5          # fp32 MIOpen on the other side: 576-term dot products


def test_thin_conv_rejects_bad_arguments():
    n = _lib()
    L = n.lib()
    x = torch.zeros(1, 1, 8, 6, device='cuda')
    w = torch.zeros(16, 1, 3, 3, device='cuda')
    b = torch.zeros(16, device='cuda')
    y = torch.zeros(1, 16, 8, 6, device='cuda')
    assert L.tai_conv_cin1_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 16, 8, 6, 3, 1, None) != 0  # W % 4
    assert L.tai_conv_cin1_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 16, 8, 8, 7, 1, None) != 0  # k
    assert L.tai_conv_cin1_forward(None, w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 16, 8, 8, 3, 1, None) != 0
    assert L.tai_conv_cout1_3x3_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 16, 8, 6, 0, None) != 0
    assert b'conv_cout1' in L.tai_sepconv_last_error()


@pytest.mark.parametrize('k', [3, 5])
@pytest.mark.parametrize('shape', [(2, 64, 32, 32), (3, 16, 20, 36)])
def test_cin1_fused_maxpool(k, shape):
    from video_frame_inpainting_amd.conv_ops import conv_bias_act, conv_bias_act_maxpool
    N, Co, H, W = shape
    g = torch.Generator().manual_seed(k + H)
    x = torch.randn(N, 1, H, W, generator=g).cuda()
    w = (torch.randn(Co, 1, k, k, generator=g) * 0.3).cuda()
    b = torch.randn(Co, generator=g).cuda()
    with torch.no_grad():
        y, yp = conv_bias_act_maxpool(x, w, b, k // 2, 'relu')
        assert torch.equal(y, conv_bias_act(x, w, b, k // 2, 'relu'))
        assert torch.equal(yp, F.max_pool2d(y, 2))


@pytest.mark.parametrize('origin', [(2, 3), (3, 4), (0, 0)])
def test_cin1_pool_into_a_halo_plane(origin):
    """The pooled output written into a larger plane at an (odd or even) origin; everything outside the window untouched."""
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, Co, H, W, k = 2, 16, 16, 24, 5
    oy, ox = origin
    ph, pw = H // 2 + oy + 3, W // 2 + ox + 5
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, 1, H, W, generator=g).cuda()
    w = torch.randn(Co, 1, k, k, generator=g).cuda() * 0.2
    b = torch.randn(Co, generator=g).cuda()
    y = torch.empty(N, Co, H, W, device='cuda')
    plane = torch.full((N, Co, ph, pw), 7.0, device='cuda')
    _native.check(L.tai_conv_cin1_forward_maxpool_window(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), plane.data_ptr(),
                                                        N, Co, H, W, k, 1, ph, pw, oy, ox, torch.cuda.current_stream().cuda_stream), 'cin1')
    ref = torch.relu(F.conv2d(x, w, b, padding=k // 2))
    assert float((y - ref).abs().max()) <= 1e-5
    assert torch.equal(plane[:, :, oy:oy + H // 2, ox:ox + W // 2], F.max_pool2d(y, 2))
    mask = torch.ones_like(plane, dtype=torch.bool)
    mask[:, :, oy:oy + H // 2, ox:ox + W // 2] = False
    assert bool((plane[mask] == 7.0).all())


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('paths', ['both', 'pooled_only', 'full_only'])
def test_activation_pool_training_form_matches_aten(relu, paths):
    """_ActPool2x2 (tai_act_maxpool2x2_forward / _backward) against relu + max_pool2d under autograd, on values with many
    exact ties (quantised, and zeros after the ReLU): the pooled gradient must go to the FIRST maximum of a window."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(3)
    z = (torch.randint(-3, 4, (3, 5, 12, 24), generator=g).float() * 0.5).cuda().requires_grad_(True)
    y, yp = conv_ops._ActPool2x2.apply(z, relu)
    zr = z.detach().clone().requires_grad_(True)
    yr = torch.relu(zr) if relu else zr
    ypr = F.max_pool2d(yr, 2)
    assert torch.equal(y, yr) and torch.equal(yp, ypr)
    gy = torch.randn(y.shape, generator=g).cuda()
    gyp = torch.randn(yp.shape, generator=g).cuda()
    loss = lambda a, b: ((a * gy).sum() if paths != 'pooled_only' else 0) + ((b * gyp).sum() if paths != 'full_only' else 0)
    loss(y, yp).backward()
    loss(yr, ypr).backward()
    assert torch.allclose(z.grad, zr.grad, rtol=0, atol=1e-6), float((z.grad - zr.grad).abs().max())


def test_conv_pool_pair_under_autograd_uses_the_fused_form():
    from video_frame_inpainting_amd import conv_ops
    torch.manual_seed(5)
    x = torch.randn(8, 64, 32, 32, device='cuda', requires_grad=True)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    y, yp = conv_ops.conv_bias_act_maxpool(x, conv.weight, conv.bias, 1, 'relu')
    assert type(yp.grad_fn).__name__ == '_ActPool2x2Backward'
    ref = torch.relu(F.conv2d(x, conv.weight, conv.bias, padding=1))
    assert float((y - ref).abs().max()) <= 1e-4 and float((yp - F.max_pool2d(ref, 2)).abs().max()) <= 1e-4
    (y.mean() + yp.mean()).backward()
    assert conv.weight.grad is not None and x.grad is not None


@pytest.mark.parametrize('k,act', [(5, 'relu'), (3, 'relu'), (3, None)])
def test_one_input_channel_layer_training_form(k, act):
    """nn.Conv2d(1, gf, k) (+ ReLU) under autograd: forward on tai_conv_cin1_forward, weight / bias gradients from
    tai_thin_conv_wrw, against fp64 autograd of conv2d."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(k)
    N, Co, H, W = 6, 64, 24, 40
    x = torch.randn(N, 1, H, W, generator=g).cuda().requires_grad_(True)     # (a generated frame difference needs its gradient)
    w = (torch.randn(Co, 1, k, k, generator=g) * 0.2).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True)
    go = torch.randn(N, Co, H, W, generator=g).cuda()
    y = conv_ops.conv_bias_act(x, w, b, k // 2, act)
    assert type(y.grad_fn).__name__ == '_ThinInConvBackward'
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = F.conv2d(xd, wd, bd, padding=k // 2)
    yd = yd * (y.detach() > 0) if act == 'relu' else yd
    rx, rw, rb = torch.autograd.grad(yd, (xd, wd, bd), go.double())
    assert float((y.double() - yd).abs().max()) <= 1e-5
    for got, ref in ((gx, rx), (gw, rw), (gb, rb)):
        assert float((got.double() - ref).abs().max()) <= 1e-4 * (1 + float(ref.abs().max()))
    assert torch.equal(torch.autograd.grad(conv_ops.conv_bias_act(x, w, b, k // 2, act), w, go)[0], gw)      # reproducible


@pytest.mark.parametrize('transposed', [True, False])
def test_one_output_channel_layer_training_form(transposed):
    """nn.ConvTranspose2d(gf, 1, 3, padding=1) + Tanh (mcnet.py:223-224) under autograd: input, weight and bias gradients
    against fp64 autograd of the reference ops."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(9)
    N, Ci, H, W = 5, 64, 20, 36
    x = torch.randn(N, Ci, H, W, generator=g).cuda().requires_grad_(True)
    wshape = (Ci, 1, 3, 3) if transposed else (1, Ci, 3, 3)
    w = (torch.randn(*wshape, generator=g) * 0.1).cuda().requires_grad_(True)
    b = torch.randn(1, generator=g).cuda().requires_grad_(True)
    go = torch.randn(N, 1, H, W, generator=g).cuda()
    y = conv_ops.conv_bias_act(x, w, b, 1, 'tanh', transposed=transposed)
    assert type(y.grad_fn).__name__ == '_ThinOutConvBackward'
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = torch.tanh(F.conv_transpose2d(xd, wd, bd, padding=1) if transposed else F.conv2d(xd, wd, bd, padding=1))
    rx, rw, rb = torch.autograd.grad(yd, (xd, wd, bd), go.double())
    assert float((y.double() - yd).abs().max()) <= 1e-5
    for got, ref in ((gx, rx), (gw, rw), (gb, rb)):
        assert float((got.double() - ref).abs().max()) <= 1e-4 * (1 + float(ref.abs().max()))


def test_training_entry_points_reject_bad_arguments():
    """Every training entry of the C ABI validates before it launches: null pointers, shapes it does not take."""
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    x = torch.zeros(1, 8, 8, 16, device='cuda')
    p = x.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    assert L.tai_thin_conv_wrw(p, p, p, p, p, 1, 8, 8, 16, 4, s) != 0                     # k not in {3, 5}
    assert L.tai_thin_conv_wrw(p, p, None, None, p, 1, 8, 8, 16, 3, s) != 0               # neither output wanted
    assert L.tai_thin_conv_wrw(p, p, p, p, p, 1, 8, 8, 18, 3, s) != 0                     # W % 4
    assert L.tai_act_maxpool2x2_forward(p, p, p, 8, 7, 16, 1, s) != 0                     # odd H
    assert L.tai_act_maxpool2x2_backward(None, None, p, None, 8, 8, 16, 1, s) != 0        # no output
    assert L.tai_window_scale_bias_lrelu(p, p, p, 1, 1, 8, 6, 0.2, s) != 0                # HW % 4
    assert L.tai_window_scale_lrelu_backward(p, p, p, None, p, 1, 1, 8, 128, 0.2, s) != 0
    assert L.tai_convlstm_gates_backward(p, p, p, None, None, p, p, 1, 2, 128, 1.0, s) != 0   # no incoming gradient
    assert L.tai_sn_power_iteration(p, p, p, None, 8, 16, 0, s) != 0                      # Ip < 1
    assert L.tai_upsample_bilinear2x_backward(None, p, 1, 4, 4, s) != 0
    assert L.tai_conv_cout1_5x5_forward(p, p, None, p, 1, 8, 8, 18, s) != 0               # W % 4
    assert L.tai_conv3x3_wino_wrw_window(p, p, p, None, p, 1, 8, 8, 8, 16, 8, 16, 1, 2, s) != 0   # window outside the plane
    assert b'window' in L.tai_sepconv_last_error() or len(L.tai_sepconv_last_error()) > 0
    torch.cuda.synchronize()
