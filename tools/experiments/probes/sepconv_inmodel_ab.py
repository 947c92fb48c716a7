#!/usr/bin/env python3
"""Same-box, in-model A/B of the one-tile kernel (18) and the persistent kernel (20): the full bi-TAI forward (configs[1]) captured
twice, once with each, and replayed alternately; run under `rocprofv3 --kernel-trace` the two kernels' durations inside the
replays can be read side by side (fwd::sepconv_forward_ab<5, 0> grid 655360 vs fwd::sepconv_forward_persistent).
Usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/sepconv_inmodel_ab.py [variants, default 18,20]   (e.g. 23,26: the persistent
kernel with its type-A waves alternating 2, 2, 0 against a constant 1; the instantiations differ in their last template argument)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd import separable_convolution as sc
from video_frame_inpainting_amd.graph import GraphedForward

VARS = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '18,20').split(','))
dev = torch.device('cuda:0')
model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, 5, 5, 5))
graphs = {}
with torch.no_grad():
    model(5, P, Fo)
    for var in VARS:
        sc.set_forward_variant(var)
        graphs[var] = GraphedForward(model, 5, P, Fo, warmup=1)
    sc.set_forward_variant(0)
    for rnd in range(10):
        for var in VARS:
            graphs[var]()
    torch.cuda.synchronize()
    for var in VARS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graphs[var]()
        e1.record(); torch.cuda.synchronize()
        print('kernel %d: %.3f ms per replayed forward' % (var, e0.elapsed_time(e1) / 5))
