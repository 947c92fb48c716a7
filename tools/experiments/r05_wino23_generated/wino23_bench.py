#!/usr/bin/env python3
"""The generated eight-wave F(2x2, 3x3) kernel (csrc/wino23_conv.hip.inc) against the compiler-scheduled one (csrc/wino_conv.hip.inc) on the
bi-TAI layers that keep that arithmetic, same process, alternating rounds.  Usage: python tools/wino23_bench.py [rounds]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native

L = _native.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
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


shapes = [(64, 64, 64, 128, 128), (64, 128, 64, 128, 128), (64, 64, 128, 64, 64), (64, 128, 64, 64, 64), (32, 256, 64, 64, 64), (64, 256, 128, 32, 32),
          (64, 512, 256, 16, 16), (64, 256, 128, 16, 16), (64, 128, 256, 16, 16), (32, 64, 64, 64, 64), (160, 512, 512, 4, 4), (160, 256, 256, 8, 8),
          (16, 64, 64, 256, 256), (16, 3, 64, 256, 256)]
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + C + K)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** .5).cuda()
    b = torch.randn(K, generator=g).cuda()
    U0 = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U0.data_ptr(), K, C, s), 'transform')
    U1 = torch.empty(L.tai_conv3x3_wino23_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino23_transform_weights(w.data_ptr(), U1.data_ptr(), K, C, s), 'transform')
    y0, y1 = torch.empty((N, K, H, W), device='cuda'), torch.empty((N, K, H, W), device='cuda')
    xs = (ctypes.c_void_p * 1)(x.data_ptr())
    old = lambda: _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U0.data_ptr(), b.data_ptr(), y0.data_ptr(), N, C, K, H, W, 1, s), 'old')
    new = lambda: _native.check(L.tai_conv3x3_wino23_forward_ex(xs, 1, U1.data_ptr(), b.data_ptr(), y1.data_ptr(), None, None, None, N, C, K, H, W, 1, s), 'new')
    old(); new()
    err = float((y1 - y0).abs().max()) / float(y0.abs().max())
    fl = 2.0 * N * K * C * 9 * H * W
    for rnd in range(rounds):
        t0, t1 = timed(old), timed(new)
        print('x(%d,%d,%d,%d)->%d round %d  generated %.1f us (%.0f TF direct, %.3f of the fp32 MFMA peak)  compiler-scheduled %.1f us (%.0f TF)  ratio %.3f  | difference %.1e'
              % (N, C, H, W, K, rnd, t1, fl / t1 / 1e6, fl / 2.25 / t1 / 1e6 / 157.3, t0, fl / t0 / 1e6, t0 / t1, err), flush=True)
