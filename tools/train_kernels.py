#!/usr/bin/env python3
"""Device time per kernel of one configs[2] training update (TAI_gray, 32 clips), torch.profiler.  Usage: python tools/train_kernels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import create_training_environment
vfi.configure_miopen()
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
model = vfi.create_model('TAI_gray')
env = create_training_environment(model, 1, '/tmp/ckpt_bench', 'x', 5, 5, 5, [128, 128], 1.0, 0.02, 1e-4, 0.5, 64, 3, 3, [0, 0], device=dev)
env.sync_replicas()
clips = torch.from_numpy(synthetic.make_clips(32, 15, 1, 128, 128, 1003))
def step():
    env.K, env.T, env.F = 5, 5, 5
    env.train(); env.train_step(clips[:, :5], clips[:, 10:], clips[:, 5:10])
for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key[:110], e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type == torch.autograd.DeviceType.CUDA]
rows.sort(key=lambda r: -r[2])
print('total kernel ms %.2f' % sum(r[2] for r in rows))
for r in rows[:40]:
    print('  %-110s %5d %9.3f ms' % r)
