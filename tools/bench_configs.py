#!/usr/bin/env python3
"""Secondary BASELINE.json configs on one GPU (not bench lines): cfg4 = TAI_color 256x256 RGB K=F=3 T=5 batch 16,
cfg5 = TAI_gray 128x128 T=10 batch 32; hipGraph replay, frames/s and the sepconv forward duration at that shape."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd import separable_convolution as sc
from video_frame_inpainting_amd.graph import GraphedForward

dev = torch.device('cuda:0')
torch.backends.cudnn.allow_tf32 = False
for name, key, B, C, H, W, K, T, F in (('cfg4', 'TAI_color', 16, 3, 256, 256, 3, 5, 3), ('cfg5', 'TAI_gray', 32, 1, 128, 128, 5, 10, 5)):
    m = synthetic.seeded_init(vfi.create_model(key), 0); m.to(dev).eval()
    clips = synthetic.make_clips(B, K + T + F, C, H, W, synthetic.SEEDS[name])
    P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, K, T, F))
    t0 = time.time()
    g = GraphedForward(m, T, P, Fo, warmup=1)
    print('[%s] captured after %.1f s' % (name, time.time() - t0), flush=True)
    for _ in range(2): g()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
    for _ in range(n): g()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ks = 51
    gen = torch.Generator().manual_seed(7)
    inp = (torch.rand(B, C, H + ks - 1, W + ks - 1, generator=gen) * 2 - 1).to(dev)
    v = (torch.randn(B, ks, H, W, generator=gen) * 0.1).to(dev); h = (torch.randn(B, ks, H, W, generator=gen) * 0.1).to(dev)
    with torch.no_grad():
        for _ in range(5): vfi.SeparableConvolution.apply(inp, v, h, ks)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): vfi.SeparableConvolution.apply(inp, v, h, ks)
        e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nb = sc.forward_bytes(B, C, H, W, ks)
    flops = 2.0 * B * C * H * W * (ks * ks + ks)
    print(json.dumps({'config': name, 'model': key, 'clips': B, 'ms_per_step': round(dt * 1e3, 2), 'frames_per_s': round(B * T / dt, 1),
                      'sepconv_us': round(us, 1), 'sepconv_TBps_algorithmic': round(nb / us / 1e6, 2), 'sepconv_frac_hbm': round(nb / us / 1e6 / 8, 3),
                      'sepconv_TFLOPs_factored': round(flops / us / 1e6, 1), 'sepconv_frac_fp32_vector_peak': round(flops / us / 1e6 / 157.3, 3)}), flush=True)
    del m, g
    torch.cuda.empty_cache()
