#!/usr/bin/env python3
"""Same-box A/B of the configs[1] forward with the default F(2x2, 3x3) convolutions and with F(4x4, 3x3) on the wide layers
(conv_ops.set_winograd_tile(4)): both captured as hipGraphs, replayed alternately, HIP events around groups of replays; under
`rocprofv3 --kernel-trace --stats` the per-kernel totals of the two graphs can be read side by side.
Usage: python tools/wino43_inmodel_ab.py [min_workgroups]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops, synthetic
from video_frame_inpainting_amd.graph import GraphedForward

if len(sys.argv) > 1:
    conv_ops.WINO43_MIN_WORKGROUPS = int(sys.argv[1])
dev = torch.device('cuda:0')
model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, 5, 5, 5))
graphs = {}
with torch.no_grad():
    model(5, P, Fo)
    for tile in (2, 4):
        conv_ops.set_winograd_tile(tile)
        model(5, P, Fo)
        graphs[tile] = GraphedForward(model, 5, P, Fo, warmup=1)
    conv_ops.set_winograd_tile(2)
    for rnd in range(3):
        for tile in (2, 4):
            graphs[tile]()
    torch.cuda.synchronize()
    for rnd in range(3):
        for tile in (2, 4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                graphs[tile]()
            e1.record(); torch.cuda.synchronize()
            print('round %d tile %d: %.3f ms per replayed forward' % (rnd, tile, e0.elapsed_time(e1) / 5), flush=True)
