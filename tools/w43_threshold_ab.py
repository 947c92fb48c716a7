#!/usr/bin/env python3
"""Same-box A/B of the configs[1] forward over conv_ops.WINO43_MIN_WORKGROUPS (below it a layer stays on F(2x2, 3x3)) or, with --channels,
over conv_ops.WINO43_MIN_CHANNELS: one hipGraph per value, replayed alternately.
Usage: python tools/w43_threshold_ab.py [--channels] [values ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops, synthetic
from video_frame_inpainting_amd.graph import GraphedForward

KNOB = 'WINO43_MIN_CHANNELS' if '--channels' in sys.argv else 'WINO43_MIN_WORKGROUPS'
values = [int(v) for v in sys.argv[1:] if v != '--channels'] or ([128, 64, 32, 16] if KNOB.endswith('CHANNELS') else [400, 256, 150, 100, 64])
dev = torch.device('cuda:0')
model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, 5, 5, 5))
graphs = {}
with torch.no_grad():
    model(5, P, Fo)
    for v in values:
        setattr(conv_ops, KNOB, v)
        model(5, P, Fo)
        graphs[v] = GraphedForward(model, 5, P, Fo, warmup=1)
    for rnd in range(2):
        for v in values:
            graphs[v]()
    torch.cuda.synchronize()
    for rnd in range(3):
        row = []
        for v in values:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                graphs[v]()
            e1.record(); torch.cuda.synchronize()
            row.append('%d: %.3f ms' % (v, e0.elapsed_time(e1) / 5))
        print('round %d  ' % rnd + '   '.join(row), flush=True)
