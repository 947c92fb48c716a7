#!/usr/bin/env python3
"""Time the Winograd-domain weight-gradient kernel (tai_conv3x3_wino_wrw) against MIOpen's (aten.convolution_backward) on the
3x3 layer shapes of the bi-TAI training step (32 clips per GPU, 128 x 128)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops

SHAPES = [(32, 64, 64, 128, 128), (32, 64, 128, 64, 64), (32, 128, 128, 64, 64), (32, 128, 256, 32, 32), (32, 256, 256, 32, 32),
          (32, 512, 256, 16, 16), (32, 512, 512, 16, 16), (32, 64, 51, 128, 128), (32, 256, 128, 32, 32), (32, 128, 64, 64, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6


for N, C, K, H, W in SHAPES:
    x = torch.randn(N, C, H, W, device='cuda'); go = torch.randn(N, K, H, W, device='cuda')
    w = torch.randn(K, C, 3, 3, device='cuda')
    mi = lambda: torch.ops.aten.convolution_backward(go, x, w, [K], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    ours = lambda: conv_ops.wino_weight_grad(x, go)
    a, b = ours(), mi()
    err = (a - b).abs().max().item() / b.abs().max().item()
    t_o, t_m = timeit(ours), timeit(mi)
    L = conv_ops._native.lib()
    prev = L.tai_conv3x3_wino_wrw_set_paired(0)
    t_u = timeit(ours)
    same = torch.equal(ours(), a) if W % 32 else None
    L.tai_conv3x3_wino_wrw_set_paired(prev)
    flops = 2.0 * N * H * W * C * K * 9
    print('N%d C%d K%d %dx%d: wino wrw %7.1f us (%5.1f TF direct-equivalent; 8-tile chunks %7.1f us)   MIOpen %7.1f us   x%.2f   max rel diff %.1e' % (
        N, C, K, H, W, t_o, flops / t_o * 1e-6, t_u, t_m, t_m / t_o, err), flush=True)
