#!/usr/bin/env python3
"""Is this box's memory system the one the numbers were tuned on?  Prints, for the GPU the process sees: a 1 GiB -> 1 GiB copy
and a 1 GiB read-reduce (GB/s), the sepconv forward at B = 32 / 64 / 96 / 128 / 160 (HIP events; taps re-read by back-to-back
launches, so B = 32 is Infinity-Cache-warm and B >= 64 streams from HBM), per-sample cost, for kernels 16 and 18.
Usage: python tools/box_diag.py"""
import os
import socket
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import separable_convolution as sc

dev = torch.device('cuda:0')
KS, H, W = 51, 128, 128


def ev_time(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    print('host %s  device %s  torch %s' % (socket.gethostname(), torch.cuda.get_device_name(0), torch.__version__))
    free, total = torch.cuda.mem_get_info()
    print('memory free %.1f / %.1f GB' % (free / 1e9, total / 1e9))
    a = torch.empty(256 << 20, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    us = ev_time(lambda: b.copy_(a), 10)
    print('copy 1 GiB -> 1 GiB: %.1f us = %.2f TB/s (read + write)' % (us, 2 * a.numel() * 4 / us / 1e6))
    us = ev_time(lambda: a.sum(), 10)
    print('sum over 1 GiB:      %.1f us = %.2f TB/s (read)' % (us, a.numel() * 4 / us / 1e6))
    del a, b
    g = torch.Generator().manual_seed(7)
    N = 160
    inp = (torch.rand(N, 1, H + KS - 1, W + KS - 1, generator=g) * 2 - 1).to(dev)
    v = (torch.randn(N, KS, H, W, generator=g) * 0.1).to(dev)
    h = (torch.randn(N, KS, H, W, generator=g) * 0.1).to(dev)
    f = vfi.SeparableConvolution.apply
    with torch.no_grad():
        for variant in (16, 18):
            prev = sc.set_forward_variant(variant)
            for B in (32, 64, 96, 128, 160):
                us = ev_time(lambda: f(inp[:B], v[:B], h[:B], KS), 20)
                nb = sc.forward_bytes(B, 1, H, W, KS)
                print('kernel %d  B=%3d: %7.1f us  %.2f us per sample  %.3f of 8 TB/s' % (variant, B, us, us / B, nb / us / 1e6 / 8), flush=True)
            sc.set_forward_variant(prev)


if __name__ == '__main__':
    main()
