#!/usr/bin/env python3
"""Which ATen operators (by input shape) are left in one training update: torch.profiler grouped by op and shape."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import create_training_environment
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = vfi.create_model('TAI_gray')
env = create_training_environment(model, 1, '/tmp/ckpt_bench', 'x', 5, 5, 5, [128, 128], 1.0, 0.02, 1e-4, 0.5, 64, 3, 3, [0, 0], device=dev)
env.sync_replicas()
clips = torch.from_numpy(synthetic.make_clips(B, 15, 1, 128, 128, 1003))
def step():
    env.K, env.T, env.F = 5, 5, 5
    env.train(); env.train_step(clips[:, :5], clips[:, 10:], clips[:, 5:10])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0)
    if t > 300:
        rows.append((t / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
for t, c, k, sh in sorted(rows, reverse=True)[:45]:
    print('%7.2f ms %5d  %-34s %s' % (t, c, k[:34], sh))
