#!/usr/bin/env python3
"""Same-box, in-model A/B of the one-tile kernel (18) and the persistent kernel (20): the full bi-TAI forward (configs[1]) captured
twice, once with each, and replayed alternately; run under `rocprofv3 --kernel-trace` the two kernels' durations inside the
replays can be read side by side (fwd::sepconv_forward_ab<5, 0> grid 655360 vs fwd::sepconv_forward_persistent)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd import separable_convolution as sc
from video_frame_inpainting_amd.graph import GraphedForward

dev = torch.device('cuda:0')
model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, 5, 5, 5))
graphs = {}
with torch.no_grad():
    model(5, P, Fo)
    for var in (18, 20):
        sc.set_forward_variant(var)
        graphs[var] = GraphedForward(model, 5, P, Fo, warmup=1)
    sc.set_forward_variant(0)
    for rnd in range(6):
        for var in (18, 20):
            graphs[var]()
    torch.cuda.synchronize()
    for var in (18, 20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graphs[var]()
        e1.record(); torch.cuda.synchronize()
        print('kernel %d: %.3f ms per replayed forward' % (var, e0.elapsed_time(e1) / 5))
