"""Winograd F(2x2, 3x3) in its generated eight-wave form (csrc/wino23_conv.hip.inc, tools/gen_wino23_asm.py) against an fp64 convolution of the
same operands and against the compiler-scheduled kernel of the same arithmetic (csrc/wino_conv.hip.inc).  Reference layers:
nn.Conv2d(C, K, 3, padding=1) (+ReLU) of src/models/mcnet/mcnet.py:79-118,165-176,198-224."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
_ACT = {None: 0, 'relu': 1, 'tanh': 2}
TOL = 4e-6          # F(2x2, 3x3) in fp32 against the magnitude sum of the dot product (tests/test_gpu_wino_conv.py's bound)


def _run(x_parts, w, b, act, ypool=False, addx=None, want_y2=False):
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, cp, H, W = x_parts[0].shape
    C, K = cp * len(x_parts), w.shape[0]
    s = torch.cuda.current_stream().cuda_stream
    U = torch.empty(L.tai_conv3x3_wino23_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino23_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    yp = torch.full((N, K, H // 2, W // 2), float('nan'), device='cuda') if ypool else None
    y2 = torch.full((N, K, H, W), float('nan'), device='cuda') if want_y2 else None
    ptrs = (ctypes.c_void_p * len(x_parts))(*[p.data_ptr() for p in x_parts])
    _native.check(L.tai_conv3x3_wino23_forward_ex(ptrs, len(x_parts), U.data_ptr(), b.data_ptr(), y.data_ptr(), yp.data_ptr() if ypool else None,
                                                  addx.data_ptr() if addx is not None else None, y2.data_ptr() if want_y2 else None,
                                                  N, C, K, H, W, _ACT[act], s), 'forward_ex')
    return y, yp, y2


def _operands(N, C, K, H, W, seed=0):
    g = torch.Generator().manual_seed(seed + N + C + K + H)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    return x, w, b


# one tile, tiles that straddle images and workgroups, K not a multiple of 64, channel counts that are no multiple of 4, bi-TAI shapes
SHAPES = [(1, 4, 64, 2, 2), (2, 8, 70, 6, 10), (3, 64, 64, 8, 20), (5, 12, 3, 2, 4), (2, 64, 64, 128, 128), (4, 128, 64, 64, 64),
          (3, 51, 51, 16, 24), (2, 65, 64, 8, 12), (2, 3, 64, 16, 16), (7, 256, 130, 16, 16)]


@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
@pytest.mark.parametrize('shape', SHAPES)
def test_wino23_matches_fp64_conv(shape, act):
    x, w, b = _operands(*shape)
    got = _run([x], w, b, act)[0]
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = torch.relu(ref) if act == 'relu' else (torch.tanh(ref) if act == 'tanh' else ref)
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
    assert torch.isfinite(got).all()
    assert float(((got.double() - ref).abs() / (1 + mag)).max()) <= TOL


def test_wino23_sees_every_tap_and_the_padding():
    """exact small integers: every tap, the zero padding on all four sides, tiles that wrap around rows and images"""
    g = torch.Generator().manual_seed(2)
    x = torch.randint(-3, 4, (3, 8, 6, 10), generator=g).float().cuda()
    w = torch.randint(-2, 3, (5, 8, 3, 3), generator=g).float().cuda()
    b = torch.randint(-4, 5, (5,), generator=g).float().cuda()
    assert torch.equal(_run([x], w, b, None)[0], F.conv2d(x, w, b, padding=1))


@pytest.mark.parametrize('nparts', [2, 4])
def test_wino23_reads_channel_parts_without_a_cat(nparts):
    x, w, b = _operands(3, 64, 64, 16, 32, seed=nparts)
    parts = [p.contiguous() for p in x.chunk(nparts, dim=1)]
    assert torch.equal(_run(parts, w, b, 'relu')[0], _run([x], w, b, 'relu')[0])


@pytest.mark.parametrize('shape,nparts', [((3, 64, 64, 16, 24), 1), ((2, 64, 70, 8, 8), 2), ((4, 128, 64, 32, 32), 2)])
def test_wino23_pooled_and_unpool_add_outputs(shape, nparts):
    x, w, b = _operands(*shape)
    N, C, K, H, W = shape
    parts = [p.contiguous() for p in x.chunk(nparts, dim=1)]
    plain = _run(parts, w, b, 'relu')[0]
    y, yp, _ = _run(parts, w, b, 'relu', ypool=True)
    assert torch.equal(y, plain) and torch.equal(yp, F.max_pool2d(plain, 2))
    lin = _run(parts, w, b, None)[0]
    addx = torch.randn(N, K, H // 2, W // 2, generator=torch.Generator().manual_seed(3)).cuda()
    want = lin.clone()
    want[:, :, ::2, ::2] += addx
    y, _, y2 = _run(parts, w, b, None, addx=addx, want_y2=True)
    assert torch.equal(y, lin) and torch.equal(y2, want)
    assert torch.equal(_run(parts, w, b, None, addx=addx)[0], want)


def test_wino23_is_bit_reproducible_and_close_to_the_compiler_scheduled_kernel():
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    x, w, b = _operands(8, 64, 64, 64, 64)
    first = _run([x], w, b, 'relu')[0].clone()
    for _ in range(10):
        assert torch.equal(_run([x], w, b, 'relu')[0], first)
    s = torch.cuda.current_stream().cuda_stream
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(64, 64), device='cuda')
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), 64, 64, s), 'transform')
    old = torch.empty_like(first)
    _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), old.data_ptr(), 8, 64, 64, 64, 64, 1, s), 'forward')
    assert float((first - old).abs().max()) <= 2e-6 * float(old.abs().max())      # same products, another summation order


def test_wino23_rejects_what_it_cannot_run():
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    x, w, b = _operands(1, 8, 64, 8, 8)
    y = torch.empty(1, 64, 8, 8, device='cuda')
    U = torch.empty(L.tai_conv3x3_wino23_weight_floats(64, 8), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    xs = (ctypes.c_void_p * 2)(x.data_ptr(), x.data_ptr())
    call = lambda *a: L.tai_conv3x3_wino23_forward_ex(*a)
    assert call(xs, 1, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, None, 1, 8, 64, 7, 8, 0, s) != 0        # odd H
    assert call(xs, 2, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, None, 1, 12, 64, 8, 8, 0, s) != 0       # parts of 6 channels
    assert call(xs, 1, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, y.data_ptr(), 1, 8, 64, 8, 8, 0, s) != 0  # y2 without addx
    assert call(xs, 1, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, None, 1, 8, 64, 8, 8, 3, s) != 0        # act
    assert call(xs, 5, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, None, 1, 8, 64, 8, 8, 0, s) != 0        # nparts
