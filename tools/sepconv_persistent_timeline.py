#!/usr/bin/env python3
"""Per-round timeline of the persistent forward kernel (tools build, variant 120 = kernel 20 writing 100 MHz stamps): where a type-A
and a type-B wave spend each round -- behind 30 back-to-back MFMA convolutions in a graph (the state the in-model launch finds) and
from idle.  Usage: TAI_NATIVE_TIMING_LIB=1 python tools/sepconv_persistent_timeline.py"""
import os
import sys
os.environ['TAI_NATIVE_TIMING_LIB'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_frame_inpainting_amd import _native, conv_ops

dev = torch.device('cuda:0')
L = _native.lib()
KS, H, W, B = 51, 128, 128, 160
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 120        # 120: the shipped scheme (= 126 beyond the Infinity Cache); 123: type A alternating 2, 2, 0; 125 / 126 / 127: type A at constant priority 0 / 1 / 2


def sep(inp, v, h, out, variant):
    L.tai_sepconv_set_forward_variant(variant)
    _native.check(L.tai_sepconv_forward(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), B, 1, H, W, KS,
                                        torch.cuda.current_stream().cuda_stream), 'fwd')
    L.tai_sepconv_set_forward_variant(0)


g = torch.Generator().manual_seed(7)
inp = (torch.rand(B, 1, H + KS - 1, W + KS - 1, generator=g) * 2 - 1).to(dev)
v = (torch.randn(B, KS, H, W, generator=g) * 0.1).to(dev)
h = (torch.randn(B, KS, H, W, generator=g) * 0.1).to(dev)
out = torch.zeros(B, 1, H, W, device=dev)
x = torch.randn(64, 256, 32, 32, generator=g).to(dev)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev)
b = torch.zeros(256, device=dev)
with torch.no_grad():
    for _ in range(3):
        conv_ops.conv_bias_act(x, w, b, 1, 'relu'); sep(inp, v, h, out, VARIANT)
    torch.cuda.synchronize()
    NCONV = int(sys.argv[2]) if len(sys.argv) > 2 else 30     # ~0.33 ms of fp32 MFMA each
    for name, nconv in (('behind %d convolutions (%.0f ms of MFMA)' % (NCONV, NCONV / 3.0), NCONV), ('from idle', 0)):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(nconv):
                conv_ops.conv_bias_act(x, w, b, 1, 'relu')
            sep(inp, v, h, out, VARIANT)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        r = out.view(torch.int64).reshape(-1)[:256 * 8 * 8 * 8].cpu().numpy().reshape(256, 8, 8, 8)[:, :, :5, :6].astype(np.int64)
        t0 = r[:, :, 0, 0].min()
        us = (r - t0) / 100.0
        print('== %s: kernel span %.1f us (first round start -> last stamp)' % (name, us.max()))
        A, Bw = us[:, 4:], us[:, :4]
        print('type A waves (mean over 256 workgroups x 4 waves), us since kernel start:')
        print('  round   start  taps issued  taps+patch there  row loop done   | wait for taps  row loop')
        for rd in range(5):
            m = A[:, :, rd].mean(axis=(0, 1))
            print('  %d     %7.1f   %7.1f        %7.1f         %7.1f       |   %6.1f     %6.1f' % (rd, m[0], m[1], m[2], m[3], m[2] - m[0], m[3] - m[2]))
        print('type B waves:')
        print('  round   start  patch ready  row loop done  next patch issued  fold done  announced | row loop   fold   tail')
        for rd in range(5):
            m = Bw[:, :, rd].mean(axis=(0, 1))
            print('  %d     %7.1f  %7.1f      %7.1f        %7.1f          %7.1f   %7.1f   | %6.1f  %6.1f  %6.1f' % (
                rd, m[0], m[1], m[2], m[3], m[4], m[5], m[2] - m[1], m[4] - m[3], m[5] - m[4]))
        ends = us[:, :, 4, :].max(axis=(1, 2))
        print('workgroup end: p50 %.1f  p90 %.1f  max %.1f' % (np.percentile(ends, 50), np.percentile(ends, 90), ends.max()))
        del graph
