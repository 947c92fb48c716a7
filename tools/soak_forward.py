#!/usr/bin/env python3
"""Soak of the replayed bi-TAI forward (configs[1], the persistent sepconv kernel inside): N replays, every output compared bit for
bit with the first replay's (the forward is deterministic: fixed summation orders everywhere).  A rare stale read behind the
LDS-counter synchronisation would show as a mismatch; a hang as the watchdog of the caller.  Usage: python tools/soak_forward.py [N] [intree] [color] [long] [split]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.graph import GraphedForward

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
if len(sys.argv) > 2 and sys.argv[2] == 'intree':        # every 3x3 layer on the in-tree Winograd kernel (no MIOpen layer left)
    from video_frame_inpainting_amd import conv_ops
    conv_ops.WINO_MIN_WORKGROUPS = 0
if 'split' in sys.argv[2:]:                 # the opt-in split-bf16 Winograd arithmetic (csrc/wino_split.hip.inc: eight waves, two barriers per chunk)
    from video_frame_inpainting_amd import conv_ops
    conv_ops.set_winograd_arithmetic('bf16x3')
dev = torch.device('cuda:0')
COLOR = 'color' in sys.argv[2:]              # configs[3]: TAI_color 256 x 256, 16 clips (three-channel sepconv kernel 19)
LONG = 'long' in sys.argv[2:]                # configs[4]: T = 10 (640 tiles... 320 samples per sepconv launch: three rounds and more)
model = synthetic.seeded_init(vfi.create_model('TAI_color' if COLOR else 'TAI_gray'), 0).to(dev).eval()
K, T, F = (3, 5, 3) if COLOR else ((5, 10, 5) if LONG else (5, 5, 5))
clips = synthetic.make_clips(16 if COLOR else 32, K + T + F, 3 if COLOR else 1, 256 if COLOR else 128, 256 if COLOR else 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, K, T, F))
with torch.no_grad():
    g = GraphedForward(model, T, P, Fo, warmup=1)
    first = {k: v.clone() for k, v in g().items()}
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    for i in range(N):
        out = g()
        same = all(torch.equal(out[k], first[k]) for k in first)
        if not same:
            bad += 1
            if bad <= 5:
                print('replay %d differs: %s' % (i, {k: (float((out[k] - first[k]).abs().max()), bool(torch.isnan(out[k]).any())) for k in first}), flush=True)
        if (i + 1) % 200 == 0:
            torch.cuda.synchronize()
            print('%d replays, %d mismatches, %.1f s' % (i + 1, bad, time.time() - t0), flush=True)
print('done: %d replays, %d mismatches' % (N, bad))
sys.exit(1 if bad else 0)
