#!/usr/bin/env python3
"""Ablations of the generated F(4x4, 3x3) chunk loop (tools build: TAI_NATIVE_TIMING_LIB=1; results wrong by design), same process,
alternating with the full kernel.  Usage: TAI_NATIVE_TIMING_LIB=1 python tools/w43_ablate.py [N,C,K,H,W]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TAI_NATIVE_TIMING_LIB', '1')
import torch
from video_frame_inpainting_amd import _native

L = _native.lib()
N, C, K, H, W = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (64, 256, 256, 32, 32)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
x = torch.randn(N, C, H, W, generator=g).cuda()
w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** .5).cuda()
b = torch.randn(K, generator=g).cuda()
U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, C), device='cuda')
_native.check(L.tai_conv3x3_wino43_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
y = torch.empty((N, K, H, W), device='cuda')
names = {0: 'full kernel', 101: 'without the patch transform arithmetic', 102: 'without the patch loads', 103: 'without the weight DMA',
         104: 'without the barrier', 105: 'without the MFMAs', 106: 'without transform, patch loads and DMA', 107: '... and without the barrier',
         108: '... and without the operand reads (MFMAs alone)', 109: 'without transform and patch loads',
         110: 'SCHEDULE transform in the middle, loads / DMA behind the groups', 111: 'SCHEDULE transform first, loads / DMA behind the groups',
         112: 'SCHEDULE transform in the middle, loads / DMA in front'}      # (the full kernel: transform first, loads / DMA in front)
if len(sys.argv) > 2:
    names = {k: v for k, v in names.items() if k in (0, 110, 111, 112)}


def timed(form, n=20):
    assert L.tai_conv3x3_wino43_set_waves(form) >= 0
    for _ in range(3):
        _native.check(L.tai_conv3x3_wino43_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s), 'forward')
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        L.tai_conv3x3_wino43_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


print('x(%d,%d,%d,%d)->%d' % (N, C, H, W, K))
for rnd in range(3):
    print('round %d: ' % rnd + '; '.join('%s %.1f us' % (names[f], timed(f)) for f in sorted(names)), flush=True)
L.tai_conv3x3_wino43_set_waves(0)
