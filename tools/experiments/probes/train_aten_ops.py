#!/usr/bin/env python3
"""Which ATen operators (not in-tree kernels) the training update spends device time in, by operator and input shapes:
torch.profiler with record_shapes on one update of configs[2]'s per-GPU share.  Usage: python tools/train_aten_ops.py [rows]"""
import os, sys, tempfile, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import create_training_environment
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(sys.stderr):
    env = create_training_environment(vfi.create_model('TAI_gray'), 1, tempfile.mkdtemp(), 'x', 5, 5, 5, [128, 128], 1.0, 0.02, 1e-4, 0.5, 64, 3, 3, [0, 0], device=dev)
env.sync_replicas()
clips = torch.from_numpy(synthetic.make_clips(32, 15, 1, 128, 128, 1003))
def step():
    env.K, env.T, env.F = 5, 5, 5
    env.train(); env.train_step(clips[:, :5], clips[:, 10:], clips[:, 5:10])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and (getattr(e, 'self_device_time_total', 0) or 0) > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print('aten operators with device time of their own: %.2f ms in one update' % (tot / 1e3))
for e in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 45]:
    print('%-28s calls %4d  self device %7.2f ms  shapes %s' % (e.key, e.count, e.self_device_time_total / 1e3, str(e.input_shapes)[:150]))
