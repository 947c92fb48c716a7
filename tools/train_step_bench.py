#!/usr/bin/env python3
"""Time the cfg3 training step (TAI_gray, K=T=F=5, 128x128, clips/GPU = 32, GAN + reconstruction losses, Adam) on one GPU."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import create_training_environment

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
if os.environ.get('TAI_WINOGRAD_ARITHMETIC'):       # 'bf16x3': the opt-in split-bf16 Winograd GEMMs (forward and input gradients)
    from video_frame_inpainting_amd import conv_ops
    conv_ops.set_winograd_arithmetic(os.environ['TAI_WINOGRAD_ARITHMETIC'])
if os.environ.get('TAI_PARTS_AUTOGRAD') == '0':       # A/B: convolutions of channel parts concatenate first under autograd (until round 4)
    from video_frame_inpainting_amd import conv_ops as _co
    _co.PARTS_UNDER_AUTOGRAD = False
if os.environ.get('TAI_WINO43_AUTOGRAD') == '0':     # A/B: everything under autograd on F(2x2, 3x3) (rounds 2-4)
    from video_frame_inpainting_amd import conv_ops as _c43
    _c43.WINO43_UNDER_AUTOGRAD = False
if os.environ.get('TAI_DISC_S2D') == '0':           # A/B: the discriminator's 4x4 stride-2 layers on MIOpen (until round 5)
    from video_frame_inpainting_amd import sn_discriminator as _snd
    _snd._s2d_applies = lambda *a: False
if os.environ.get('TAI_WRW_TILE'):                   # A/B: 2 = the F(2x2, 3x3)-domain weight-gradient kernel everywhere (rounds 2-5)
    from video_frame_inpainting_amd import conv_ops as _cw
    _cw.set_weight_gradient_tile(int(os.environ['TAI_WRW_TILE']))
torch.manual_seed(0); np.random.seed(0)
model = vfi.create_model('TAI_gray')
env = create_training_environment(model, 1, '/tmp/ckpt_bench', 'x', 5, 5, 5, [128, 128], 1.0, 0.02, 1e-4, 0.5, 64, 3, 3, [0, 0], device=dev,
                                  graph_step=os.environ.get('TAI_GRAPH_STEP') == '1')
env.sync_replicas()
clips = torch.from_numpy(synthetic.make_clips(B, 15, 1, 128, 128, 1003))
def step():
    env.K, env.T, env.F = 5, 5, 5
    env.train(); env.train_step(clips[:, :5], clips[:, 10:], clips[:, 5:10])
t0 = time.time(); step(); torch.cuda.synchronize(); print('[train] first step %.1f s' % (time.time() - t0), flush=True)
for _ in range(3):      # with TAI_GRAPH_STEP=1: the second eager warm-up, the capture, a first replay
    step()
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print('[train] B=%d: %.1f ms/step -> %.1f clips/s, %.1f GB peak' % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)
print(env.get_current_errors())
if len(sys.argv) > 3:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=int(os.environ.get('TAI_PROF_ROWS', '18')), max_name_column_width=90))
