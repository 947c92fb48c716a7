#!/usr/bin/env python3
"""Per-layer timing of every convolution on the bi-TAI forward (TAI_gray, clips/GPU = 32): which MIOpen solver each
shape gets and how far from the fp32 MFMA peak it runs.  Hooks the real model, so shapes are exactly the path's."""
import collections, sys, time, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = vfi.create_model('TAI_gray'); m.apply(vfi.util.weights_init); m.to(dev).eval()
clips = synthetic.make_clips(B, 15, 1, 128, 128, 1002)
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))

shapes = collections.OrderedDict()
orig = F.conv2d
def rec(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    key = (tuple(x.shape), tuple(w.shape), padding if isinstance(padding, int) else tuple(padding))
    shapes[key] = shapes.get(key, 0) + 1
    return orig(x, w, b, stride, padding, dilation, groups)
F.conv2d = rec
torch.nn.functional.conv2d = rec
import torch.nn as nn
_orig_fwd = nn.Conv2d._conv_forward
def _cf(self, input, weight, bias):
    key = (tuple(input.shape), tuple(weight.shape), self.padding)
    shapes[key] = shapes.get(key, 0) + 1
    return _orig_fwd(self, input, weight, bias)
nn.Conv2d._conv_forward = _cf
with torch.no_grad():
    m(5, P, Fo)
torch.cuda.synchronize()
nn.Conv2d._conv_forward = _orig_fwd
F.conv2d = orig
print('%d distinct conv shapes, %d conv calls per forward' % (len(shapes), sum(shapes.values())), flush=True)
tot = 0.0
rows = []
for (xs, ws, pad), n in shapes.items():
    x = torch.randn(*xs, device=dev); w = torch.randn(*ws, device=dev) * 0.05; b = torch.zeros(ws[0], device=dev)
    p = pad if isinstance(pad, int) else pad[0]
    for _ in range(2): orig(x, w, b, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it): orig(x, w, b, 1, p)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / it
    flops = 2.0 * xs[0] * ws[0] * ws[1] * ws[2] * ws[3] * xs[2] * xs[3]
    rows.append((ms * n, ms, n, xs, ws, flops / ms / 1e9))
    tot += ms * n
    print('x%-22s w%-20s calls=%2d  %8.3f ms/call  %7.1f TFLOP/s  total %8.2f ms' % (xs, ws, n, ms, flops / ms / 1e9, ms * n), flush=True)
print('sum of conv time per forward: %.1f ms' % tot)
