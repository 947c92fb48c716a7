#!/usr/bin/env python3
"""Every conv_bias_act call of one bi-TAI forward (the layers that end in a fused max pool -- the last convolution of
every encoder stage -- are not listed, only the 5x5 / 7x7 ones among them, which fall back to conv_bias_act) (TAI_gray, clips/GPU = 32): which kernel takes it (thin / Winograd-MFMA
/ MIOpen), time per call on that path and -- for 3x3 layers -- on the other one, and the total per forward."""
import collections, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic, conv_ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = vfi.create_model('TAI_gray'); m.apply(vfi.util.weights_init); m.to(dev).eval()
clips = synthetic.make_clips(B, 15, 1, 128, 128, 1002)
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))

calls = collections.OrderedDict()
orig = conv_ops.conv_bias_act
def rec(x, weight, bias, padding, act, transposed=False):
    parts = len(x) if isinstance(x, (list, tuple)) else 1
    x0 = x[0] if parts > 1 else x
    key = ((x0.shape[0], x0.shape[1] * parts, x0.shape[2], x0.shape[3]), tuple(weight.shape), padding, act, transposed, parts)
    calls[key] = calls.get(key, 0) + 1
    return orig(x, weight, bias, padding, act, transposed)
import video_frame_inpainting_amd.mcnet as mc, video_frame_inpainting_amd.tai as tai
mc.conv_bias_act = rec; tai.conv_bias_act = rec; conv_ops_conv = conv_ops.conv_bias_act; conv_ops.conv_bias_act = rec   # (conv_bias_act_maxpool's fallback goes through it too)
with torch.no_grad():
    m(5, P, Fo)
torch.cuda.synchronize()
mc.conv_bias_act = orig; tai.conv_bias_act = orig; conv_ops.conv_bias_act = orig

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n

tot = collections.Counter(); alt_gain = 0.0
print('%d distinct shapes, %d calls per forward' % (len(calls), sum(calls.values())))
for (xs, ws, pad, act, tr, parts), n in calls.items():
    x = torch.randn(*xs, device=dev)
    if parts > 1: x = tuple(t.contiguous() for t in x.chunk(parts, dim=1))
    w = torch.randn(*ws, device=dev) * 0.05
    co = ws[1] if tr else ws[0]
    b = torch.zeros(co, device=dev)
    thr = conv_ops.WINO_MIN_WORKGROUPS
    with torch.no_grad():
        conv_ops.WINO_MIN_WORKGROUPS = thr
        y = orig(x, w, b, pad, act, tr)
        path = 'wino' if ('wino', tr) in getattr(w, '_tai_derived', {}) else ('thin' if (ws[0] == 1 or ws[1] == 1) else 'miopen')
        ms = t(lambda: orig(x, w, b, pad, act, tr))
        other = float('nan')
        if ws[2] == 3 and path != 'thin' and xs[2] % 2 == 0 and ws[1 if not tr else 0] >= 8:
            conv_ops.WINO_MIN_WORKGROUPS = 0 if path == 'miopen' else 10 ** 9
            other = t(lambda: orig(x, w, b, pad, act, tr))
            conv_ops.WINO_MIN_WORKGROUPS = thr
    wgs = ((xs[0] * xs[2] * xs[3] // 4 + 63) // 64) * ((co + 63) // 64)
    tot[path] += ms * n
    if other == other and other < ms: alt_gain += (ms - other) * n
    print('x%-22s w%-20s k%d %-5s parts=%d calls=%3d  %-6s %8.3f ms/call   other path %8.3f   wgs %5d  total %7.2f ms' % (xs, ws, ws[2], 'convT' if tr else '', parts, n, path, ms, other, wgs, ms * n), flush=True)
print('per forward:', {k: round(v, 2) for k, v in tot.items()}, ' possible gain by switching paths: %.2f ms' % alt_gain)
