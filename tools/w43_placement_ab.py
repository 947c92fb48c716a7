#!/usr/bin/env python3
"""Same-box A/B of the workgroup placement of the F(4x4, 3x3) kernels (tai_conv3x3_wino43_set_placement: 1 = aware of the 8 XCDs, 0 = plain
dispatch order): the configs[1] forward as one hipGraph per setting, replayed alternately; then the weight-gradient kernel per layer.
Usage: python tools/w43_placement_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import _native, conv_ops, synthetic
from video_frame_inpainting_amd.graph import GraphedForward

L = _native.lib()
dev = torch.device('cuda:0')
model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, 5, 5, 5))
graphs, outs = {}, {}
with torch.no_grad():
    model(5, P, Fo)
    for v in (0, 1):
        L.tai_conv3x3_wino43_set_placement(v)
        model(5, P, Fo)
        graphs[v] = GraphedForward(model, 5, P, Fo, warmup=1)
        outs[v] = {k: t.clone() for k, t in graphs[v]().items()}
    assert all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0]), 'the placement changed the results'
    for rnd in range(2):
        for v in (0, 1):
            graphs[v]()
    torch.cuda.synchronize()
    for rnd in range(3):
        row = []
        for v in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                graphs[v]()
            e1.record(); torch.cuda.synchronize()
            row.append('%s: %.3f ms' % ('XCD-aware' if v else 'dispatch order', e0.elapsed_time(e1) / 5))
        print('forward, round %d  ' % rnd + '   '.join(row), flush=True)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (N, C, K, H, W) in [(64, 64, 64, 128, 128), (64, 128, 128, 64, 64), (64, 256, 256, 32, 32), (64, 512, 1024, 16, 16), (160, 64, 64, 64, 64)]:
    x = torch.randn(N, C, H, W).cuda(); go = torch.randn(N, K, H, W).cuda()
    t = {}
    for v in (0, 1):
        L.tai_conv3x3_wino43_set_placement(v)
        t[v] = timed(lambda: conv_ops.wino_weight_grad(x, go, with_bias=True))
    print('weight gradient x(%d,%d,%d,%d)->%d  dispatch order %.1f us  XCD-aware %.1f us' % (N, C, H, W, K, t[0], t[1]), flush=True)
L.tai_conv3x3_wino43_set_placement(1)
