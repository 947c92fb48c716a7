#!/usr/bin/env python3
"""Where the weight-gradient time of one configs[2] training update goes: every tai_conv3x3_wino_wrw / _window call bracketed by HIP events,
grouped by shape, under both workgroup placements (tai_conv3x3_wino43_set_placement).  Usage: python tools/train_wrw_times.py"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import _native, synthetic
from video_frame_inpainting_amd.environments import create_training_environment

vfi.configure_miopen()
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
model = vfi.create_model('TAI_gray')
env = create_training_environment(model, 1, '/tmp/ckpt_bench', 'x', 5, 5, 5, [128, 128], 1.0, 0.02, 1e-4, 0.5, 64, 3, 3, [0, 0], device=dev)
env.sync_replicas()
clips = torch.from_numpy(synthetic.make_clips(32, 15, 1, 128, 128, 1003))
L = _native.lib()
records = []


def wrap(name):
    fn = getattr(L, name)

    def call(*args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        records.append((tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool) and abs(x) < 1 << 20), a, b))
        return rc
    return call


def step():
    env.K, env.T, env.F = 5, 5, 5
    env.train(); env.train_step(clips[:, :5], clips[:, 10:], clips[:, 5:10])


for _ in range(2):
    step()
plain = {n: getattr(L, n) for n in ('tai_conv3x3_wino_wrw', 'tai_conv3x3_wino_wrw_window')}
table = collections.defaultdict(lambda: [0, {0: 0.0, 1: 0.0}])
for v in (0, 1, 0, 1):
    L.tai_conv3x3_wino43_set_placement(v)
    for n in plain:
        setattr(L, n, wrap(n))
    del records[:]
    step()
    torch.cuda.synchronize()
    for n, f in plain.items():
        setattr(L, n, f)
    for key, a, b in records:
        table[key][0] += 1
        table[key][1][v] += a.elapsed_time(b)
L.tai_conv3x3_wino43_set_placement(1)
rows = sorted(table.items(), key=lambda kv: -kv[1][1][1])
print('%-44s %6s %12s %12s' % ('N C K H W [in_h in_w oy ox]', 'calls', 'dispatch ms', 'XCD-aware ms'))
for key, (n, t) in rows:
    print('%-44s %6d %12.3f %12.3f' % (' '.join(map(str, key)), n // 4, t[0] / 2, t[1] / 2))
print('total %.2f ms against %.2f ms' % (sum(t[0] for _, (n, t) in rows) / 2, sum(t[1] for _, (n, t) in rows) / 2))
