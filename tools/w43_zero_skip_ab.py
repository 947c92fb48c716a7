#!/usr/bin/env python3
"""Same-process A/B of the zero-tap MFMA skipping of tai_conv3x3_wino43_forward_blocks: shift_k = 7 / 5 (last blocks carry zero taps: skipped)
against shift_k = 9 / 6 on the same plane and weights (the same S x S blocks, nothing skipped; timing only -- as a 9 x 9 / 6 x 6 filter the
weights mean something else).  Usage: python tools/w43_zero_skip_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native, conv_ops

L = _native.lib()
s = torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for k, kfull, (N, Cin, K, H, W) in ((7, 9, (64, 128, 256, 32, 32)), (5, 6, (64, 64, 128, 64, 64))):
    S, top, left, in_h, in_w = conv_ops.halo_geometry(H, W, kfull)
    plane = torch.randn(N, Cin, in_h, in_w, device='cuda')
    w = torch.randn(K, Cin, k, k, device='cuda') * 0.02
    wb = conv_ops._block3x3_weight(w)
    U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, S * S * Cin), device='cuda')
    _native.check(L.tai_conv3x3_wino43_transform_weights(wb.data_ptr(), U.data_ptr(), K, S * S * Cin, s), 'transform')
    b = torch.zeros(K, device='cuda')
    y = torch.empty(N, K, H, W, device='cuda')
    run = lambda kk: _native.check(L.tai_conv3x3_wino43_forward_blocks(plane.data_ptr(), kk, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, 0, 0, 0, 0,
                                                                       N, S * S * Cin, K, H, W, in_h, in_w, 1, 2, 1, s), 'blocks')
    for rnd in range(3):
        print('%d x %d layer x(%d,%d,%d,%d)->%d round %d: skipping %.1f us   not skipping %.1f us' %
              (k, k, N, Cin, H, W, K, rnd, timed(lambda: run(k)), timed(lambda: run(kfull))), flush=True)
