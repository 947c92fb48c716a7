#!/usr/bin/env python3
"""Marginal cost of the sepconv forward launch inside a hipGraph: graph A = R x [two 51->51 Winograd convolutions that write
the taps; sepconv], graph B = R x [the two convolutions]; (time(A) - time(B)) / R is what the launch adds to a replayed
step -- the number that matters for frames/s -- next to what rocprofv3's kernel trace attributes to it.
Usage: python tools/sepconv_graph_marginal.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops

dev = torch.device('cuda:0')
KS, H, W, N, R = 51, 128, 128, 160, 8
sep = vfi.SeparableConvolution.apply


def replay_ms(graph, n=6):
    graph.replay(); graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    from video_frame_inpainting_amd import separable_convolution as sc
    if len(sys.argv) > 1:
        sc.set_forward_variant(int(sys.argv[1]))
        print('forward variant', sys.argv[1])
    g = torch.Generator().manual_seed(7)
    inp = (torch.rand(N, 1, H + KS - 1, W + KS - 1, generator=g) * 2 - 1).to(dev)
    v = torch.empty(N, KS, H, W, device=dev)
    h = torch.empty(N, KS, H, W, device=dev)
    x51 = (torch.randn(N, KS, H, W, generator=g) * 0.5).to(dev)
    w51 = (torch.randn(KS, KS, 3, 3, generator=g) / np.sqrt(KS * 9) * 0.3).to(dev)
    b51 = (torch.randn(KS, generator=g) * 0.01).to(dev)
    conv = lambda xs, out: conv_ops.conv_bias_act(xs, w51, b51, 1, None, out=out)
    outs = []

    def body(with_sep, groups):
        nb = N // groups
        for gi in range(groups):
            s = slice(gi * nb, (gi + 1) * nb)
            conv(x51[s], v[s]); conv(x51[s], h[s])
            if with_sep:
                outs.append(sep(inp[s], v[s], h[s], KS))

    with torch.no_grad():
        body(True, 1); body(True, 5)
        torch.cuda.synchronize()
        res = {}
        for name, with_sep, groups in (('convs only, 1 group', False, 1), ('convs + sepconv, 1 group of 160', True, 1),
                                       ('convs only, 5 groups', False, 5), ('convs + sepconv, 5 groups of 32', True, 5)):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(R):
                    body(with_sep, groups)
            res[name] = replay_ms(graph)
            print('%-36s %8.3f ms per replay (min %.3f) = %.1f us per repetition' % (name, res[name][0], res[name][1], res[name][0] * 1e3 / R), flush=True)
            del graph
        for a, b, tag in (('convs + sepconv, 1 group of 160', 'convs only, 1 group', 'one launch of 160'),
                          ('convs + sepconv, 5 groups of 32', 'convs only, 5 groups', 'five launches of 32 behind their convs')):
            us = (res[a][0] - res[b][0]) * 1e3 / R
            print('marginal sepconv cost, %s: %.1f us per 160 samples -> %.3f of 8 TB/s on 1,100,311,040 algorithmic bytes' % (
                tag, us, 1100311040.0 / (us * 1e-6) / 8e12), flush=True)


if __name__ == '__main__':
    main()
