#!/usr/bin/env python3
"""The three discriminator passes of one training update (generator pass; real and fake of the discriminator pass), SNDiscriminator at
configs[2]'s shape, with the 4x4 stride-2 layers on MIOpen (until round 5) and as 3x3 layers on space-to-depth planes on the in-tree Winograd
kernels (sn_discriminator._s2d_applies): ms per three passes and the kernels by device time.  Usage: python tools/disc_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import sn_discriminator as snd
vfi.configure_miopen()
dev = torch.device('cuda:0')
torch.manual_seed(0)
D = snd.SNDiscriminator((128, 128), 1, 3, 64, 3).to(dev)
fake = torch.randn(32, 15, 1, 128, 128, device=dev).tanh()
real = torch.randn(32, 15, 1, 128, 128, device=dev).tanh()
applies = snd._s2d_applies

def passes():
    f = fake.clone().requires_grad_()
    lg = D(f); F.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)).backward()
    D.zero_grad()
    lr, lf = D(real), D(fake)
    (F.binary_cross_entropy_with_logits(lr, torch.ones_like(lr)) + F.binary_cross_entropy_with_logits(lf, torch.zeros_like(lf))).backward()

for route in ('miopen', 's2d'):
    snd._s2d_applies = applies if route == 's2d' else (lambda *a: False)
    for _ in range(3):
        passes()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        passes()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.2f ms per three discriminator passes' % (route, e0.elapsed_time(e1) / 3), flush=True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        passes(); torch.cuda.synchronize()
    rows = [(e.key[:100], e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type == torch.autograd.DeviceType.CUDA]
    rows.sort(key=lambda r: -r[2])
    print('total kernel ms %.2f' % sum(r[2] for r in rows))
    for r in rows[:22]:
        print('  %-100s %4d %8.3f ms' % r)
