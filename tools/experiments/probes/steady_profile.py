#!/usr/bin/env python3
"""One steady-state bi-TAI forward (TAI_gray, 32 clips) for rocprofv3 --kernel-trace: two warm-up forwards, a 1 s pause,
then the forward to look at.  tools/steady_profile_summary.py reads the trace and sums the kernels after the pause."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = vfi.create_model('TAI_gray'); m.apply(vfi.util.weights_init); m.to(dev).eval()
clips = synthetic.make_clips(32, 15, 1, 128, 128, 1002)
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))
with torch.no_grad():
    for _ in range(2): m(5, P, Fo)
    torch.cuda.synchronize(); time.sleep(1.0)
    t0 = time.time(); m(5, P, Fo); torch.cuda.synchronize(); print('eager forward %.1f ms' % ((time.time() - t0) * 1e3))
