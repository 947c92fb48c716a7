#!/usr/bin/env python3
"""The forms of the F(4x4, 3x3) kernel against each other and against F(2x2, 3x3), same process, alternating rounds:
0 = generated chunk loop (tools/gen_wino43_asm.py), 8 / 4 = round 4's compiler-scheduled forms.
Usage: python tools/wino43_forms_bench.py [rounds]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from video_frame_inpainting_amd import _native

L = _native.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def make(which, x, w, b, act=1):
    N, C, H, W = x.shape
    K = w.shape[0]
    s = torch.cuda.current_stream().cuda_stream
    pre = 'tai_conv3x3_wino43' if which == 43 else 'tai_conv3x3_wino'
    U = torch.empty(getattr(L, pre + '_weight_floats')(K, C), device='cuda')
    _native.check(getattr(L, pre + '_transform_weights')(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    fwd = getattr(L, pre + '_forward')

    def run():
        _native.check(fwd(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, act, s), 'forward')
        return y
    return run


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


shapes = [(64, 256, 256, 32, 32), (64, 128, 128, 64, 64), (64, 512, 1024, 16, 16), (64, 128, 256, 32, 32), (64, 512, 256, 32, 32),
          (64, 256, 128, 64, 64), (160, 256, 256, 16, 16), (160, 512, 512, 8, 8), (64, 256, 256, 16, 16)]
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + C + K)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** .5).cuda()
    b = torch.randn(K, generator=g).cuda()
    r43, r23 = make(43, x, w, b), make(23, x, w, b)
    outs = {}
    for form in (0, 8):
        L.tai_conv3x3_wino43_set_waves(form)
        outs[form] = r43().clone()
    y23 = r23().clone()
    scale = float(y23.abs().max())
    err = {f: float((outs[f] - y23).abs().max()) / scale for f in outs}
    fl = 2.0 * N * K * C * 9 * H * W
    for rnd in range(rounds):
        t = {}
        for form in (0, 8):
            L.tai_conv3x3_wino43_set_waves(form)
            t[form] = timed(r43)
        t23 = timed(r23)
        print('x(%d,%d,%d,%d)->%d round %d  generated %.1f us (%.0f TF direct, %.3f of the fp32 MFMA peak)  r04 eight-wave %.1f us  F(2x2) %.1f us   '
              'generated / r04 %.3f   | vs F(2x2): generated %.2e  r04 %.2e' %
              (N, C, H, W, K, rnd, t[0], fl / t[0] / 1e6, fl / 4 / t[0] / 1e6 / 157.3, t[8], t23, t[8] / t[0], err[0], err[8]), flush=True)
L.tai_conv3x3_wino43_set_waves(0)
