#!/usr/bin/env python3
"""Kernel-time table of one eager bi-TAI forward (TAI_gray, 32 clips) from torch.profiler (rocprofv3 changes which
MIOpen solvers run, torch.profiler does not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
dev = torch.device('cuda:0')
m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0); m.to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, 1002)
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))
with torch.no_grad():
    for _ in range(2): m(5, P, Fo)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m(5, P, Fo); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
wino = sum(e.device_time_total for e in rows if 'wino::conv3x3' in e.key)
print('total device time %.2f ms in %d kernel kinds; Winograd convolution kernels %.2f ms, everything else %.2f ms'
      % (tot / 1e3, len(rows), wino / 1e3, (tot - wino) / 1e3))
for e in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 28]:
    print('%-78s calls %4d  total %7.2f ms  avg %8.1f us' % (e.key[:78], e.count, e.device_time_total / 1e3, e.device_time_total / e.count))
