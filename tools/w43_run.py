#!/usr/bin/env python3
"""Launches the F(4x4, 3x3) kernel a few times per shape and form, for rocprofv3 (tools/prof_w43.sh).
Usage: python3 tools/w43_run.py forms shapes   e.g.  0 64,256,256,32,32   (forms: 0, or 101-112 in the tools build) 64,512,1024,16,16"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native

L = _native.lib()
forms = [int(f) for f in sys.argv[1].split(',')]
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]] or [(64, 256, 256, 32, 32)]
s = torch.cuda.current_stream().cuda_stream
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + C + K)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** .5).cuda()
    b = torch.randn(K, generator=g).cuda()
    U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino43_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
    y = torch.empty((N, K, H, W), device='cuda')
    for form in forms:
        L.tai_conv3x3_wino43_set_waves(form)
        for _ in range(12):
            _native.check(L.tai_conv3x3_wino43_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s), 'forward')
        torch.cuda.synchronize()
L.tai_conv3x3_wino43_set_waves(0)
