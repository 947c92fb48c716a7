"""Parity of the HIP separable convolution with the CPU oracle, through the C ABI (the autograd op calls
tai_sepconv_forward / tai_sepconv_backward).  Tolerances (fp32): |d| <= 1e-5 (1+|ref|) forward,
2e-5 gradients; the reference is the oracle accumulated in fp64, so the bound covers ANY fp32 summation order."""
import numpy as np
import pytest
import torch

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import _native
from video_frame_inpainting_amd import separable_convolution as sc
from oracle import sepconv_oracle as so

pytestmark = pytest.mark.gpu
FWD_TOL, BWD_TOL = 1e-5, 2e-5
DEV = 'cuda:0'


def _rel(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / (1 + np.abs(b))))


def _case(B, C, H, W, ks, seed):
    g = torch.Generator().manual_seed(seed)
    inp = torch.rand(B, C, H + ks - 1, W + ks - 1, generator=g) * 2 - 1          # tanh-range image
    v = torch.randn(B, ks, H, W, generator=g) * 0.1
    h = torch.randn(B, ks, H, W, generator=g) * 0.1
    gO = torch.randn(B, C, H, W, generator=g)
    return inp, v, h, gO


def test_library_is_the_hip_build():
    L = _native.lib()
    assert L.tai_sepconv_version() >= 100


SHAPES = [
    (2, 1, 16, 128, 51),    # the fast path, gray
    (1, 3, 12, 36, 51),     # ragged tile: H % 8 != 0, W < 128, RGB
    (1, 2, 8, 132, 51),     # two column tiles, channel passes
    (1, 1, 5, 7, 51),       # W % 4 != 0 -> generic kernels
    (2, 3, 9, 10, 7),       # other kernel size -> generic kernels
    (1, 1, 1, 4, 51),       # single row
    (3, 1, 32, 32, 51),     # the golden-fixture size
    (2, 1, 40, 256, 51),    # two column tiles x three 16-row tiles (ragged last tile), default mixed kernel
]


@pytest.mark.parametrize('B,C,H,W,ks', SHAPES)
def test_forward_and_backward_match_oracle(B, C, H, W, ks):
    inp, v, h, gO = _case(B, C, H, W, ks, 11)
    di, dv, dh = (t.to(DEV).requires_grad_() for t in (inp, v, h))
    out = vfi.SeparableConvolution.apply(di, dv, dh, ks)
    out.backward(gO.to(DEV))
    ref = so.forward(inp.numpy(), v.numpy(), h.numpy(), ks, f64=True)
    rI, rV, rH = so.backward(gO.numpy(), inp.numpy(), v.numpy(), h.numpy(), ks, f64=True)
    assert _rel(out.detach().cpu().numpy(), ref) < FWD_TOL
    assert _rel(di.grad.cpu().numpy(), rI) < BWD_TOL
    assert _rel(dv.grad.cpu().numpy(), rV) < BWD_TOL
    assert _rel(dh.grad.cpu().numpy(), rH) < BWD_TOL
    # and the literal fp32 restatement of the reference's loops is within the same bound of the HIP result
    lit = so.forward(inp.numpy(), v.numpy(), h.numpy(), ks)
    assert _rel(out.detach().cpu().numpy(), lit.astype(np.float64)) < 2 * FWD_TOL


@pytest.mark.parametrize('variant', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27])
@pytest.mark.parametrize('C', [1, 3])
def test_every_forward_variant(variant, C):
    inp, v, h, _ = _case(2, C, 24, 128, 51, 5)
    ref = so.forward(inp.numpy(), v.numpy(), h.numpy(), 51, f64=True)
    prev = sc.set_forward_variant(variant)
    try:
        with torch.no_grad():
            out = vfi.SeparableConvolution.apply(inp.to(DEV), v.to(DEV), h.to(DEV), 51)
        assert _rel(out.cpu().numpy(), ref) < FWD_TOL
    finally:
        sc.set_forward_variant(prev)


@pytest.mark.parametrize('variant', [16, 18, 20])
@pytest.mark.parametrize('B,H,W', [(2, 40, 256), (1, 16, 132), (3, 16, 128), (1, 24, 4)])
def test_mixed_wave_kernels_on_ragged_tiles_and_patch_edges(variant, B, H, W):
    """The A/B kernels on several column tiles, a ragged last row tile and a narrow last column tile; delta taps at
    (50, 50) read the LAST two columns and rows of the padded frame (kernel 18 stages those two columns separately)."""
    ks = 51
    inp, v, h, _ = _case(B, 1, H, W, ks, 17)
    prev = sc.set_forward_variant(variant)
    try:
        with torch.no_grad():
            out = vfi.SeparableConvolution.apply(inp.to(DEV), v.to(DEV), h.to(DEV), ks)
            assert _rel(out.cpu().numpy(), so.forward(inp.numpy(), v.numpy(), h.numpy(), ks, f64=True)) < FWD_TOL
            dv = torch.zeros_like(v); dv[:, 50] = 1
            dh = torch.zeros_like(h); dh[:, 50] = 1
            out = vfi.SeparableConvolution.apply(inp.to(DEV), dv.to(DEV), dh.to(DEV), ks).cpu()
            assert torch.equal(out, inp[:, :, 50:50 + H, 50:50 + W])
            # LDS keeps its contents from launch to launch: a patch element read before it was staged would be the PREVIOUS
            # launch's value.  Alternate a frame with its negation (the exact negative, bit for bit): any stale read shows.
            a, b2, dvv, dhh = inp.to(DEV), (-inp).to(DEV), v.to(DEV), h.to(DEV)
            pos = vfi.SeparableConvolution.apply(a, dvv, dhh, ks)
            for _ in range(10):
                assert torch.equal(vfi.SeparableConvolution.apply(b2, dvv, dhh, ks), -pos)
                assert torch.equal(vfi.SeparableConvolution.apply(a, dvv, dhh, ks), pos)
    finally:
        sc.set_forward_variant(prev)


@pytest.mark.parametrize('B,H,W', [(34, 128, 128), (48, 64, 256), (40, 40, 128), (24, 24, 384), (8, 16, 128), (65, 128, 128), (3, 128, 128)])
def test_persistent_forward_kernel_shape_sweep(B, H, W):
    """Kernel 20 over tile grids that are ragged in rows (H = 40, 24: clamped rows in the last tile), several column tiles wide
    (W = 256, 384: only the last column tile has the two-float row tail), one round with idle workgroups, and counts the persistent
    form does not take (tiles % 8 != 0: falls back to kernel 18) -- always the same bits as kernel 16, stale-LDS alternation included."""
    ks = 51
    g = torch.Generator().manual_seed(B * H + W)
    inp = (torch.rand(B, 1, H + ks - 1, W + ks - 1, generator=g) * 2 - 1).to(DEV)
    v = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    h = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    with torch.no_grad():
        prev = sc.set_forward_variant(16)
        try:
            want = vfi.SeparableConvolution.apply(inp, v, h, ks)
            sc.set_forward_variant(20)
            for _ in range(3):
                assert torch.equal(vfi.SeparableConvolution.apply(inp, v, h, ks), want)
                assert torch.equal(vfi.SeparableConvolution.apply(-inp, v, h, ks), -want)
            sc.set_forward_variant(0)           # the default route (persistent only with more tiles than CUs)
            assert torch.equal(vfi.SeparableConvolution.apply(inp, v, h, ks), want)
        finally:
            sc.set_forward_variant(prev)


@pytest.mark.parametrize('B', [16, 40, 72])
def test_persistent_forward_kernel_matches_the_one_tile_kernels_bit_for_bit(B):
    """Kernel 20 (one persistent workgroup per CU, the next tile's patch and taps on their way while this tile computes) runs
    the same row loops as kernels 16 / 18: identical bits.  B = 16: fewer tiles than CUs (every workgroup one tile); B = 40, 72:
    320 / 576 tiles on 256 workgroups -- one to three rounds per workgroup, both patch buffers reused.  Plus an oracle slice, and
    the alternation with the exact negation (a stale patch or tap read would show)."""
    ks, H, W = 51, 128, 128
    g = torch.Generator().manual_seed(B)
    inp = (torch.rand(B, 1, H + ks - 1, W + ks - 1, generator=g) * 2 - 1).to(DEV)
    v = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    h = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    outs = {}
    with torch.no_grad():
        # 21 / 22 / 23: the persistent kernel with the default cache policy, with nt tap loads, with nt loads and the tile list
        # walked backwards (20 picks between 21 and 23 by the tap footprint)
        for variant in (16, 18, 20, 21, 22, 23, 24, 25, 26, 27):
            prev = sc.set_forward_variant(variant)
            try:
                outs[variant] = vfi.SeparableConvolution.apply(inp, v, h, ks)
                if variant >= 20:
                    neg = -inp
                    for _ in range(4):
                        assert torch.equal(vfi.SeparableConvolution.apply(neg, v, h, ks), -outs[variant])
                        assert torch.equal(vfi.SeparableConvolution.apply(inp, v, h, ks), outs[variant])
            finally:
                sc.set_forward_variant(prev)
    assert all(torch.equal(outs[k], outs[16]) for k in (18, 20, 21, 22, 23, 24, 25, 26, 27))
    sl = slice(B - 2, B)
    ref = so.forward(inp[sl].cpu().numpy(), v[sl].cpu().numpy(), h[sl].cpu().numpy(), ks, f64=True)
    assert _rel(outs[20][sl].cpu().numpy(), ref) < FWD_TOL


@pytest.mark.parametrize('B', [160, 320])
def test_persistent_forward_kernel_at_the_round_counts_the_product_runs(B):
    """The launches the product makes: configs[1] / [2] call the op with T * B = 160 samples at 128 x 128 (1,280 tiles = 5 rounds
    of 256 persistent workgroups), configs[4] (T = 10) with 320 (10 rounds).  Through the DEFAULT route (no variant forced) the
    result must carry the same bits as kernel 16 (one tile per workgroup, kernel.cu:19-47 restated per tile), an fp64 oracle
    slice of the LAST two samples and of two from the MIDDLE must agree to the forward tolerance, and alternating the frames
    with their exact negation must alternate the result exactly (LDS keeps a previous round's patch and taps: a stale read
    of either shows as a non-negated value)."""
    ks, H, W = 51, 128, 128
    g = torch.Generator().manual_seed(1000 + B)
    inp = (torch.rand(B, 1, H + ks - 1, W + ks - 1, generator=g) * 2 - 1).to(DEV)
    v = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    h = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    L = _native.lib()
    assert L.tai_sepconv_default_forward_variant(1, W, ks) == 20     # the persistent kernel IS the default; 1,280 / 2,560 tiles > 256 CUs and % 8 == 0
    with torch.no_grad():
        prev = sc.set_forward_variant(16)
        try:
            want = vfi.SeparableConvolution.apply(inp, v, h, ks)
            sc.set_forward_variant(0)
            got = vfi.SeparableConvolution.apply(inp, v, h, ks)
            assert torch.equal(got, want)
            neg = -inp
            for _ in range(3):
                assert torch.equal(vfi.SeparableConvolution.apply(neg, v, h, ks), -want)
                assert torch.equal(vfi.SeparableConvolution.apply(inp, v, h, ks), want)
        finally:
            sc.set_forward_variant(prev)
    for sl in (slice(B - 2, B), slice(B // 2 - 1, B // 2 + 1)):
        ref = so.forward(inp[sl].cpu().numpy(), v[sl].cpu().numpy(), h[sl].cpu().numpy(), ks, f64=True)
        assert _rel(got[sl].cpu().numpy(), ref) < FWD_TOL


@pytest.mark.parametrize('i,j', [(0, 0), (50, 50), (25, 3), (1, 48), (49, 2)])
def test_delta_taps_shift_exactly(i, j):
    B, C, H, W, ks = 1, 1, 16, 128, 51
    g = torch.Generator().manual_seed(3)
    inp = torch.randn(B, C, H + ks - 1, W + ks - 1, generator=g)
    v = torch.zeros(B, ks, H, W); v[:, i] = 1
    h = torch.zeros(B, ks, H, W); h[:, j] = 1
    with torch.no_grad():
        out = vfi.SeparableConvolution.apply(inp.to(DEV), v.to(DEV), h.to(DEV), ks).cpu()
    assert torch.equal(out, inp[:, :, i:i + H, j:j + W])          # bit-exact: every other product is an exact zero


def test_box_taps_and_linearity_at_full_size():
    # BASELINE cfg2 shape: too large for the oracle in a test, so use properties that do not depend on size
    B, C, H, W, ks = 32, 1, 128, 128, 51
    g = torch.Generator().manual_seed(8)
    inp = (torch.rand(B, C, H + ks - 1, W + ks - 1, generator=g) * 2 - 1).to(DEV)
    v = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    h = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(DEV)
    with torch.no_grad():
        ones = torch.ones_like(v)
        box = vfi.SeparableConvolution.apply(inp, ones, ones, ks)
        ref = torch.nn.functional.avg_pool2d(inp.double(), ks, stride=1) * ks * ks
        # 2601 O(1) terms that largely cancel: the fp32 bound scales with sum|terms| (~1300 here), not with |ref|
        mag = torch.nn.functional.avg_pool2d(inp.double().abs(), ks, stride=1) * ks * ks
        assert float(((box.double() - ref).abs() / mag).max()) < 2e-7
        a = vfi.SeparableConvolution.apply(inp, v, h, ks)
        b = vfi.SeparableConvolution.apply(inp, 2 * v, h, ks)           # linear in v (exact: power-of-two scale)
        assert torch.equal(b, 2 * a)
        c = vfi.SeparableConvolution.apply(0.5 * inp, v, 4 * h, ks)       # trilinear
        assert torch.equal(c, 2 * a)
        # a sample of pixels against the oracle on a 2-clip slice of the same tensors
        sl = so.forward(inp[:2].cpu().numpy(), v[:2].cpu().numpy(), h[:2].cpu().numpy(), ks, f64=True)
        assert _rel(a[:2].cpu().numpy(), sl) < FWD_TOL


def test_adjoint_identity_at_full_size():
    B, C, H, W, ks = 8, 1, 128, 128, 51
    g = torch.Generator().manual_seed(9)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    inp, v, h = mk(B, C, H + ks - 1, W + ks - 1).requires_grad_(), (mk(B, ks, H, W) * 0.1).requires_grad_(), (mk(B, ks, H, W) * 0.1).requires_grad_()
    gO = mk(B, C, H, W)
    out = vfi.SeparableConvolution.apply(inp, v, h, ks)
    out.backward(gO)
    d_in, d_v, d_h = mk(*inp.shape), mk(*v.shape) * 0.1, mk(*h.shape) * 0.1
    with torch.no_grad():
        dot = lambda a, b: float((a.double() * b.double()).sum())
        f = vfi.SeparableConvolution.apply
        for lhs, rhs in ((dot(gO, f(d_in, v.detach(), h.detach(), ks)), dot(inp.grad, d_in)),
                         (dot(gO, f(inp.detach(), d_v, h.detach(), ks)), dot(v.grad, d_v)),
                         (dot(gO, f(inp.detach(), v.detach(), d_h, ks)), dot(h.grad, d_h))):
            assert abs(lhs - rhs) <= 1e-4 * (1 + abs(lhs)), (lhs, rhs)


def _grad_input(inp, v, h, gO, variant, taps_too=True):
    L = _native.lib()
    prev = L.tai_sepconv_set_grad_input_variant(variant)
    try:
        di = inp.to(DEV).requires_grad_()
        dv, dh = v.to(DEV), h.to(DEV)
        if taps_too:
            dv.requires_grad_(); dh.requires_grad_()
        vfi.SeparableConvolution.apply(di, dv, dh, 51).backward(gO.to(DEV))
        return di.grad.clone()
    finally:
        L.tai_sepconv_set_grad_input_variant(prev)


@pytest.mark.parametrize('B,C,H,W', [(2, 3, 20, 132), (2, 1, 33, 128), (1, 1, 128, 128), (1, 3, 7, 260)])
def test_all_grad_input_kernels_match_oracle(B, C, H, W):
    # 0 / 3: wave-private strips (row loop in generated assembly) + fixed-order slab sum (default when tileable); 4: the same
    # with the HIP C++ row loop; 1: the reference-style gather; 2: the round-1 LDS row-scatter with atomics.  All against the
    # fp64 oracle, and against each other.
    inp, v, h, gO = _case(B, C, H, W, 51, 12)
    rI, _, _ = so.backward(gO.numpy(), inp.numpy(), v.numpy(), h.numpy(), 51, f64=True)
    got = {}
    for variant in (0, 1, 2, 3, 4):
        got[variant] = _grad_input(inp, v, h, gO, variant).cpu().numpy()
        assert _rel(got[variant], rI) < BWD_TOL, variant
    assert np.array_equal(got[0], got[3])
    # with no tap gradient requested there is no buffer to borrow for the tile slabs: the strips flush with atomics
    alone = _grad_input(inp, v, h, gO, 0, taps_too=False).cpu().numpy()
    assert _rel(alone, rI) < BWD_TOL


def test_default_grad_input_is_bit_reproducible():
    inp, v, h, gO = _case(4, 1, 64, 128, 51, 14)
    a = _grad_input(inp, v, h, gO, 0)
    for _ in range(3):
        assert torch.equal(a, _grad_input(inp, v, h, gO, 0))


def test_grad_input_of_delta_taps_is_an_exact_scatter():
    # v = e_i, h = e_j: out[y, x] = in[y+i, x+j], so gI[y+i, x+j] = gO[y, x] and zero elsewhere -- exactly
    B, C, H, W, ks, i, j = 2, 1, 24, 128, 51, 7, 40
    g = torch.Generator().manual_seed(15)
    inp = torch.randn(B, C, H + ks - 1, W + ks - 1, generator=g)
    gO = torch.randn(B, C, H, W, generator=g)
    v = torch.zeros(B, ks, H, W); v[:, i] = 1
    h = torch.zeros(B, ks, H, W); h[:, j] = 1
    want = torch.zeros_like(inp)
    want[:, :, i:i + H, j:j + W] = gO
    assert torch.equal(_grad_input(inp, v, h, gO, 0).cpu(), want)


def test_fused_and_separate_tap_gradient_kernels_agree():
    inp, v, h, gO = _case(2, 1, 24, 128, 51, 13)
    _, rV, rH = so.backward(gO.numpy(), inp.numpy(), v.numpy(), h.numpy(), 51, f64=True)
    L = _native.lib()
    got = {}
    for variant in (0, 1, 2, 3, 4):
        prev = L.tai_sepconv_set_grad_taps_variant(variant)
        try:
            dv, dh = v.to(DEV).requires_grad_(), h.to(DEV).requires_grad_()
            vfi.SeparableConvolution.apply(inp.to(DEV), dv, dh, 51).backward(gO.to(DEV))
        finally:
            L.tai_sepconv_set_grad_taps_variant(prev)
        assert _rel(dv.grad.cpu().numpy(), rV) < BWD_TOL and _rel(dh.grad.cpu().numpy(), rH) < BWD_TOL
        got[variant] = (dv.grad.clone(), dh.grad.clone())
    # 3 / 4 differ from 0 only in the wave priority of the gV waves: the same bits
    for variant in (3, 4):
        assert torch.equal(got[variant][0], got[0][0]) and torch.equal(got[variant][1], got[0][1])


def test_partial_gradients_and_error_reporting():
    inp, v, h, gO = _case(1, 1, 8, 128, 51, 4)
    di, dv, dh = inp.to(DEV), v.to(DEV).requires_grad_(), h.to(DEV)       # only gV requested
    vfi.SeparableConvolution.apply(di, dv, dh, 51).backward(gO.to(DEV))
    _, rV, _ = so.backward(gO.numpy(), inp.numpy(), v.numpy(), h.numpy(), 51, f64=True)
    assert _rel(dv.grad.cpu().numpy(), rV) < BWD_TOL
    L = _native.lib()
    assert L.tai_sepconv_forward(None, None, None, None, 1, 1, 8, 8, 51, None) == -1      # TAI_SEPCONV_EINVAL
    assert b'null' in L.tai_sepconv_last_error()
    assert L.tai_sepconv_forward(di.data_ptr(), dv.data_ptr(), dh.data_ptr(), di.data_ptr(), 0, 1, 8, 8, 51, None) == -1


def test_index_space_limit_is_rejected_before_any_launch():
    # the kernels decode flat 32-bit indices like the reference (SeparableConvolution_kernel.cu:35-38): >= 2^31 elements
    # in any operand is refused up front instead of wrapping around
    L = _native.lib()
    t = torch.zeros(16, device=DEV)
    p = t.data_ptr()
    assert L.tai_sepconv_forward(p, p, p, p, 4096, 1, 1024, 1024, 51, None) == -1        # B*ks*H*W = 2.2e11
    assert b'dimension' in L.tai_sepconv_last_error()
    assert L.tai_sepconv_backward(p, p, p, p, p, p, p, 1, 1, 30000, 30000, 51, None) == -1
    assert L.tai_sepconv_forward(p, p, p, p, 1, 1, 8, 8, 0, None) == -1                    # ks <= 0
    torch.cuda.synchronize()


def test_runs_on_a_side_stream_and_inside_a_graph():
    inp, v, h, _ = _case(2, 1, 16, 128, 51, 6)
    di, dv, dh = inp.to(DEV), v.to(DEV), h.to(DEV)
    ref = so.forward(inp.numpy(), v.numpy(), h.numpy(), 51, f64=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        out = vfi.SeparableConvolution.apply(di, dv, dh, 51)
    s.synchronize()
    assert _rel(out.cpu().numpy(), ref) < FWD_TOL
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        gout = vfi.SeparableConvolution.apply(di, dv, dh, 51)
    di.mul_(0.5)                      # replay must read the CURRENT contents of the captured buffers
    graph.replay()
    torch.cuda.synchronize()
    assert _rel(gout.cpu().numpy(), 0.5 * ref) < FWD_TOL
