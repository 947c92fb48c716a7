"""A/B of one model switch on the SAME box: hipGraph replays of the configs[1] forward with the attribute off / on,
alternating.  Usage: python tools/ab_forward.py merge_per_step [generator.keep_res1 ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.graph import GraphedForward
dev = torch.device('cuda:0')
m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, 1002)
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))


def setattr_path(path, value):
    obj = m
    parts = path.split('.')
    for p in parts[:-1]:
        obj = getattr(obj, p)
    setattr(obj, parts[-1], value)


for attr in sys.argv[1:]:
    graphs = {}
    for value in (False, True):
        setattr_path(attr, value)
        graphs[value] = GraphedForward(m, 5, P, Fo, warmup=1)
    setattr_path(attr, True)
    res = {False: [], True: []}
    for rep in range(4):
        for value in (False, True):
            g = graphs[value]
            g(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                g()
            torch.cuda.synchronize()
            res[value].append((time.perf_counter() - t0) / 10 * 1e3)
    print('%s: off %s ms | on %s ms' % (attr, ' '.join('%.2f' % x for x in res[False]), ' '.join('%.2f' % x for x in res[True])))
