#!/usr/bin/env python3
"""Launches the weight-gradient kernel a few times per shape, for rocprofv3 (tools/prof_wrw43.sh).  Usage: python3 tools/wrw43_run.py N,C,K,H,W ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native, conv_ops
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (N, C, K, H, W) in shapes:
    x = torch.randn(N, C, H, W).cuda(); go = torch.randn(N, K, H, W).cuda()
    for _ in range(6):
        conv_ops.wino_weight_grad(x, go, with_bias=True)
    torch.cuda.synchronize()
