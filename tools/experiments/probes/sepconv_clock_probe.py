#!/usr/bin/env python3
"""What clock does the chip hold INSIDE the sepconv forward kernel, and how long does the launch take, as a function of
what ran before it?  Uses the tools build of the library (forward variant 109 = the default A/B kernel writing, per wave,
100 MHz wall-clock stamps and shader-clock stamps instead of pixels).

  idle       the launch after 100 ms of nothing
  conv_ms    the launch at the end of a hipGraph holding ~X ms of back-to-back Winograd-MFMA convolutions (the state the
             in-model launch finds: it is the last kernel of a 63 ms forward)
  self       the launch behind 20 launches of itself
Each case for [32,1,128,128] (grid 131072, one workgroup per CU) and [160,1,128,128] (grid 655360, the in-model launch).

Usage: TAI_NATIVE_TIMING_LIB=1 python tools/sepconv_clock_probe.py"""
import os
import sys
import time
os.environ['TAI_NATIVE_TIMING_LIB'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_frame_inpainting_amd import _native, conv_ops

dev = torch.device('cuda:0')
L = _native.lib()
KS, H, W = 51, 128, 128


def sep(inp, v, h, out, variant):
    L.tai_sepconv_set_forward_variant(variant)
    _native.check(L.tai_sepconv_forward(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), inp.shape[0], 1, H, W, KS,
                                        torch.cuda.current_stream().cuda_stream), 'sepconv_forward')
    L.tai_sepconv_set_forward_variant(0)


def read_stamps(out, B):
    nblk = B * (H // 16)
    r = out.view(torch.int64).reshape(-1)[:nblk * 8 * 8].cpu().numpy().reshape(nblk * 8, 8).astype(np.uint64)
    t0, t3, c0, c1 = r[:, 0].astype(np.int64), r[:, 3].astype(np.int64), r[:, 5].astype(np.int64), r[:, 6].astype(np.int64)
    span_us = (t3.max() - t0.min()) / 100.0
    ghz = ((c1 - c0) / np.maximum(t3 - t0, 1)) * 0.1          # cycles per 10 ns tick -> GHz
    life = (t3 - t0) / 100.0
    return span_us, float(np.median(ghz)), float(ghz.min()), float(ghz.max()), float(np.median(life))


def main():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(64, 256, 32, 32, generator=g).to(dev)
    w = (torch.randn(256, 256, 3, 3, generator=g) * (2.0 / (9 * 256)) ** 0.5).to(dev)
    b = torch.zeros(256, device=dev)
    with torch.no_grad():
        conv = lambda: conv_ops.conv_bias_act(x, w, b, 1, 'relu')
        conv(); conv()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv()
        e1.record(); torch.cuda.synchronize()
        conv_ms = e0.elapsed_time(e1) / 20
        print('one (64,256->256,32x32) Winograd convolution: %.3f ms' % conv_ms)
        for B in (32, 160):
            inp = (torch.rand(B, 1, H + KS - 1, W + KS - 1, generator=g) * 2 - 1).to(dev)
            v = (torch.randn(B, KS, H, W, generator=g) * 0.1).to(dev)
            h = (torch.randn(B, KS, H, W, generator=g) * 0.1).to(dev)
            out = torch.zeros(B, 1, H, W, device=dev)
            out2 = torch.zeros(B, 1, H, W, device=dev)
            for _ in range(3):
                sep(inp, v, h, out, 109)
            torch.cuda.synchronize()
            cases = [('idle 100 ms before', 0, 0)] + [('after %3d convs (~%.0f ms of MFMA)' % (n, n * conv_ms), n, 0) for n in (4, 30, 200)] + \
                    [('after 20 launches of itself', 0, 20)]
            for name, nconv, nself in cases:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(nconv):
                        conv()
                    for _ in range(nself):
                        sep(inp, v, h, out2, 16)
                    sep(inp, v, h, out, 109)
                res = []
                for rep in range(6):
                    torch.cuda.synchronize()
                    if nconv == 0 and nself == 0:
                        time.sleep(0.1)
                    graph.replay()
                    torch.cuda.synchronize()
                    res.append(read_stamps(out, B))
                res = res[1:]
                span = np.median([r[0] for r in res]); ghz = np.median([r[1] for r in res])
                print('[%3d,1,128,128] %-36s kernel span %7.1f us (min %.1f max %.1f)  shader clock in the waves: median %.3f GHz '
                      '(min %.3f max %.3f)  wave lifetime %.1f us' % (B, name, span, min(r[0] for r in res), max(r[0] for r in res), ghz,
                                                                     min(r[2] for r in res), max(r[3] for r in res), np.median([r[4] for r in res])), flush=True)
                del graph


if __name__ == '__main__':
    main()
