#!/usr/bin/env python3
"""Same-process A/B of forward kernels (default 16 vs 18): (a) [32,1,128,128] as bench.py times it (hipGraph of 50 launches,
replayed, steady state), (b) the in-model launch [160,1,128,128] as the marginal cost inside a graph of
R x [two tap-producing 51->51 convolutions; sepconv], alternating the kernels three times.
Usage: python tools/sepconv_ab_variants.py [variants, default 16,18]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops
from video_frame_inpainting_amd import separable_convolution as sc

dev = torch.device('cuda:0')
KS, H, W, R = 51, 128, 128, 8
sep = vfi.SeparableConvolution.apply
VARIANTS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '16,18').split(',')]


def replay_us(graph, n=5):
    graph.replay(); graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    g = torch.Generator().manual_seed(7)
    N = 160
    inp = (torch.rand(N, 1, H + KS - 1, W + KS - 1, generator=g) * 2 - 1).to(dev)
    v = (torch.randn(N, KS, H, W, generator=g) * 0.1).to(dev)
    h = (torch.randn(N, KS, H, W, generator=g) * 0.1).to(dev)
    x51 = (torch.randn(N, KS, H, W, generator=g) * 0.5).to(dev)
    w51 = (torch.randn(KS, KS, 3, 3, generator=g) / np.sqrt(KS * 9) * 0.3).to(dev)
    b51 = (torch.randn(KS, generator=g) * 0.01).to(dev)
    keep = []
    with torch.no_grad():
        for var in VARIANTS:
            sc.set_forward_variant(var); keep.append(sep(inp, v, h, KS)); keep.append(sep(inp[:32], v[:32], h[:32], KS))
        conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=v); conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=h)
        torch.cuda.synchronize()
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb):
            for _ in range(R):
                conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=v); conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=h)
        graphs = {}
        for var in VARIANTS:
            sc.set_forward_variant(var)
            g32 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g32):
                for _ in range(50):
                    keep.append(sep(inp[:32], v[:32], h[:32], KS))
            g160 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g160):
                for _ in range(R):
                    conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=v); conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=h)
                    keep.append(sep(inp, v, h, KS))
            gbb = torch.cuda.CUDAGraph()            # (c) the big launch back to back, nothing in between: 1.07 GB of taps per launch, all from HBM
            with torch.cuda.graph(gbb):
                for _ in range(R):
                    keep.append(sep(inp, v, h, KS))
            graphs[var] = (g32, g160, gbb)
        # (d) as bench.py times the in-model launch: eager, the two producing convolutions then the sepconv, HIP events around the sepconv alone
        for rnd in range(2):
            for var in VARIANTS:
                sc.set_forward_variant(var)
                pairs = []
                for rep in range(13):
                    conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=v); conv_ops.conv_bias_act(x51, w51, b51, 1, None, out=h)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); keep.append(sep(inp, v, h, KS)); e1.record()
                    pairs.append((e0, e1))
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b2) * 1e3 for a, b2 in pairs[3:])
                print('events %d kernel %d: in-model [160,1,128,128] behind its two convolutions: mean %.1f us (min %.1f, max %.1f) = %.3f' % (
                    rnd, var, float(np.mean(ts)), ts[0], ts[-1], 1100311040.0 / float(np.mean(ts)) / 8e6), flush=True)
                del keep[-13:]
        sc.set_forward_variant(0)
        for rnd in range(3):
            base = replay_us(gb)
            for var in VARIANTS:
                g32, g160, gbb = graphs[var]
                us32 = replay_us(g32) / 50
                us160 = (replay_us(g160) - base) / R
                usbb = replay_us(gbb) / R
                print('round %d kernel %d: [32,1,128,128] %.2f us (%.3f)   in-model [160,1,128,128] marginal %.1f us (%.3f)   back to back %.1f us (%.3f)' % (
                    rnd, var, us32, 220062208.0 / us32 / 8e6, us160, 1100311040.0 / us160 / 8e6, usbb, 1100311040.0 / usbb / 8e6), flush=True)


if __name__ == '__main__':
    main()
