"""Pins oracle/sepconv_oracle.c (the restatement of SeparableConvolution_kernel.cu:19-162).

The reference holds no known-answer vectors for its sepconv op (it has no tests), so the oracle is
pinned analytically: delta taps, box taps, a separable Gaussian against F.conv2d, the adjoint
identity between forward and the three gradients, and a finite-difference check of the gradients.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sepconv_oracle as so


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def _case(B, C, H, W, ks, seed=0):
    inp = _rand((B, C, H + ks - 1, W + ks - 1), seed)
    v = _rand((B, ks, H, W), seed + 1, 0.3)
    h = _rand((B, ks, H, W), seed + 2, 0.3)
    return inp, v, h


@pytest.mark.parametrize('ks,i,j', [(5, 0, 0), (5, 4, 2), (51, 25, 25), (51, 50, 0), (51, 7, 44)])
def test_delta_taps_shift(ks, i, j):
    # v = e_i, h = e_j  =>  out[y,x] = in[y+i, x+j]   (.cu:40-44)
    B, C, H, W = 2, 3, 6, 9
    inp = _rand((B, C, H + ks - 1, W + ks - 1), 3)
    v = np.zeros((B, ks, H, W), np.float32); v[:, i] = 1
    h = np.zeros((B, ks, H, W), np.float32); h[:, j] = 1
    out = so.forward(inp, v, h, ks)
    np.testing.assert_array_equal(out, inp[:, :, i:i + H, j:j + W])


def test_box_taps():
    ks, B, C, H, W = 7, 1, 2, 5, 4
    inp = _rand((B, C, H + ks - 1, W + ks - 1), 4)
    ones = np.ones((B, ks, H, W), np.float32)
    out = so.forward(inp, ones, ones, ks, f64=True)
    ref = F.avg_pool2d(torch.from_numpy(inp).double(), ks, stride=1).numpy() * ks * ks
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)


def test_separable_gaussian_matches_conv2d():
    ks, B, C, H, W = 51, 1, 1, 8, 8
    inp = _rand((B, C, H + ks - 1, W + ks - 1), 5)
    g = np.exp(-0.5 * ((np.arange(ks) - 25) / 6.0) ** 2); g = (g / g.sum()).astype(np.float32)
    v = np.broadcast_to(g[None, :, None, None], (B, ks, H, W)).copy()
    out = so.forward(inp, v, v, ks)
    k2 = torch.from_numpy(np.outer(g, g).astype(np.float32))[None, None]
    ref = F.conv2d(torch.from_numpy(inp), k2).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)


def test_f32_matches_f64_within_tolerance():
    inp, v, h = _case(2, 3, 8, 8, 51, 6)
    a = so.forward(inp, v, h, 51)
    b = so.forward(inp, v, h, 51, f64=True)
    assert np.max(np.abs(a - b) / (1 + np.abs(b))) < 1e-5


@pytest.mark.parametrize('B,C,H,W,ks', [(1, 1, 4, 5, 3), (2, 3, 6, 5, 7), (1, 2, 5, 4, 51)])
def test_adjoint_identity(B, C, H, W, ks):
    # forward is trilinear in (in, v, h):  <gO, fwd(d_in, v, h)> = <gI, d_in>, and likewise for v, h.
    inp, v, h = _case(B, C, H, W, ks, 7)
    gO = _rand((B, C, H, W), 10)
    gI, gV, gH = so.backward(gO, inp, v, h, ks, f64=True)
    d_in, d_v, d_h = _rand(inp.shape, 11), _rand(v.shape, 12), _rand(h.shape, 13)
    dot = lambda a, b: float(np.sum(a.astype(np.float64) * b.astype(np.float64)))
    lhs_i = dot(gO, so.forward(d_in, v, h, ks, f64=True))
    lhs_v = dot(gO, so.forward(inp, d_v, h, ks, f64=True))
    lhs_h = dot(gO, so.forward(inp, v, d_h, ks, f64=True))
    assert abs(lhs_i - dot(gI, d_in)) <= 1e-4 * (1 + abs(lhs_i))
    assert abs(lhs_v - dot(gV, d_v)) <= 1e-4 * (1 + abs(lhs_v))
    assert abs(lhs_h - dot(gH, d_h)) <= 1e-4 * (1 + abs(lhs_h))


def test_grads_match_autograd_of_unfold_formulation():
    # Independent formulation in torch (unfold + einsum) differentiated by autograd, fp64.
    B, C, H, W, ks = 2, 2, 5, 6, 5
    inp, v, h = _case(B, C, H, W, ks, 20)
    gO = _rand((B, C, H, W), 23)
    ti = torch.from_numpy(inp).double().requires_grad_()
    tv = torch.from_numpy(v).double().requires_grad_()
    th = torch.from_numpy(h).double().requires_grad_()
    patches = F.unfold(ti.reshape(B * C, 1, H + ks - 1, W + ks - 1), ks).reshape(B, C, ks, ks, H, W)
    out = torch.einsum('bcijyx,biyx,bjyx->bcyx', patches, tv, th)
    np.testing.assert_allclose(so.forward(inp, v, h, ks, f64=True), out.detach().numpy(), rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(gO).double())
    gI, gV, gH = so.backward(gO, inp, v, h, ks, f64=True)
    np.testing.assert_allclose(gI, ti.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gV, tv.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gH, th.grad.numpy(), rtol=1e-5, atol=1e-6)
    gI32, gV32, gH32 = so.backward(gO, inp, v, h, ks)
    for a, b in ((gI32, gI), (gV32, gV), (gH32, gH)):
        assert np.max(np.abs(a - b) / (1 + np.abs(b))) < 2e-5


def test_grad_i_border_is_partial_sum():
    # gI at the padded corner (0,0) only receives tap (0,0) of output pixel (0,0)   (.cu:146-158)
    B, C, H, W, ks = 1, 1, 4, 4, 5
    inp, v, h = _case(B, C, H, W, ks, 30)
    gO = _rand((B, C, H, W), 31)
    gI, _, _ = so.backward(gO, inp, v, h, ks)
    np.testing.assert_allclose(gI[0, 0, 0, 0], gO[0, 0, 0, 0] * v[0, 0, 0, 0] * h[0, 0, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(gI[0, 0, -1, -1], gO[0, 0, -1, -1] * v[0, -1, -1, -1] * h[0, -1, -1, -1], rtol=1e-6)


def test_thread_count_does_not_change_results():
    inp, v, h = _case(2, 2, 6, 6, 9, 40)
    n = so.num_threads()
    so.set_num_threads(1); a = so.forward(inp, v, h, 9)
    so.set_num_threads(max(n, 2)); b = so.forward(inp, v, h, 9)
    so.set_num_threads(n)
    np.testing.assert_array_equal(a, b)
