#!/usr/bin/env python3
"""The weight-gradient kernel in the F(4x4, 3x3) domain (csrc/wino43_conv.hip.inc, conv3x3_wrw_gen) against the F(2x2, 3x3) kernel
(csrc/wino_wrw.hip.inc): error of both against float64 autograd on small and bi-TAI shapes, then us per call alternating in one process
(tai_conv3x3_wino_wrw_set_tile).  Usage: python tools/wrw43_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from video_frame_inpainting_amd import _native, conv_ops
L = _native.lib()
def ref64(x, go):
    xd = x.double()
    wd = torch.zeros(go.shape[1], x.shape[1], 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    return torch.autograd.grad(F.conv2d(xd, wd, None, padding=1), wd, go.double())[0]
shapes = [(1, 8, 8, 4, 16), (1, 32, 64, 4, 16), (2, 16, 16, 8, 16), (3, 20, 51, 12, 32), (1, 65, 64, 16, 16), (5, 13, 70, 8, 48), (7, 9, 3, 4, 16),
          (2, 64, 64, 128, 128), (4, 512, 128, 16, 16), (3, 128, 130, 32, 32), (32, 64, 64, 64, 64), (64, 256, 256, 32, 32)]
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    go = torch.randn(N, K, H, W, generator=g).cuda()
    res = {}
    for tile in (2, 4):
        assert L.tai_conv3x3_wino_wrw_set_tile(tile) >= 0
        dw, db = conv_ops.wino_weight_grad(x, go, with_bias=True)
        torch.cuda.synchronize()
        res[tile] = (dw, db)
    L.tai_conv3x3_wino_wrw_set_tile(2)
    if N * C * H * W <= 64 * 64 * 64 * 64:
        r = ref64(x, go)
    else:
        r = res[2][0].double()
    scale = (N * H * W) ** 0.5
    e2 = float((res[2][0].double() - r).abs().max()) / scale
    e4 = float((res[4][0].double() - r).abs().max()) / scale
    rb = go.double().sum((0, 2, 3))
    eb = float((res[4][1].double() - rb).abs().max()) / scale
    print('%s  F(2x2) %.2e  F(4x4) %.2e  bias %.2e  %s' % ((N, C, K, H, W), e2, e4, eb, 'OK' if e4 < 2e-5 and eb < 2e-5 else 'BAD'), flush=True)
# timing
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (N, C, K, H, W) in [(64, 64, 64, 128, 128), (64, 128, 128, 64, 64), (64, 256, 256, 32, 32), (64, 512, 256, 32, 32), (64, 512, 1024, 16, 16), (160, 64, 64, 64, 64), (160, 256, 256, 16, 16), (416, 256, 128, 32, 32)]:
    x = torch.randn(N, C, H, W).cuda(); go = torch.randn(N, K, H, W).cuda()
    t = {}
    for tile in (2, 4):
        L.tai_conv3x3_wino_wrw_set_tile(tile)
        t[tile] = timed(lambda: conv_ops.wino_weight_grad(x, go, with_bias=True))
    L.tai_conv3x3_wino_wrw_set_tile(2)
    print('%s  F(2x2) %.1f us  F(4x4) %.1f us  ratio %.2f' % ((N, C, K, H, W), t[2], t[4], t[2] / t[4]), flush=True)
