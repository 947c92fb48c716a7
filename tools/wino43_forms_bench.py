#!/usr/bin/env python3
"""The F(4x4, 3x3) kernel (generated chunk loop, tools/gen_wino43_asm.py) against F(2x2, 3x3), same process, alternating rounds, with the
error of each against a float64 convolution where that is cheap.  (Until the middle of round 5 this also timed round 4's compiler-scheduled
forms; they are no longer in the library -- profiles/r05_wino43_generated_ablation.txt holds that comparison.)
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
    y43, y23 = r43().clone(), r23().clone()
    n = min(N, 4)
    ref = torch.relu(F.conv2d(x[:n].double(), w.double(), b.double(), padding=1))
    mag = float(ref.abs().max())
    e43, e23 = float((y43[:n].double() - ref).abs().max()) / mag, float((y23[:n].double() - ref).abs().max()) / mag
    fl = 2.0 * N * K * C * 9 * H * W
    for rnd in range(rounds):
        t43, t23 = timed(r43), timed(r23)
        print('x(%d,%d,%d,%d)->%d round %d  F(4x4) %.1f us (%.0f TF direct, %.3f of the fp32 MFMA peak)  F(2x2) %.1f us   F(2x2) / F(4x4) %.3f   '
              '| max error / max |y| against float64: F(4x4) %.2e  F(2x2) %.2e' %
              (N, C, H, W, K, rnd, t43, fl / t43 / 1e6, fl / 4 / t43 / 1e6 / 157.3, t23, t23 / t43, e43, e23), flush=True)
