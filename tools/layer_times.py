#!/usr/bin/env python3
"""Where the configs[1] forward spends its time, per library entry point and shape: every tai_* call of one eager bi-TAI forward
(TAI_gray, 32 clips) is bracketed by HIP events; calls are grouped by (entry point, integer arguments) and listed by total time with
the direct-convolution TFLOP/s of the 3x3 layers.  Usage: python tools/layer_times.py [clips] [winograd tile 2|4]"""
import collections
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import _native, conv_ops, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if len(sys.argv) > 2:
    conv_ops.set_winograd_tile(int(sys.argv[2]))
dev = torch.device('cuda:0')
m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
clips = synthetic.make_clips(B, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))
L = _native.lib()
records = []


def wrap(name):
    fn = getattr(L, name)

    def call(*args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        records.append((name, tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool) and abs(x) < 1 << 20), a, b))
        return rc
    return call


names = [n for n in _native.declared_symbols() if n.startswith(('tai_conv', 'tai_sepconv_forward', 'tai_upsample', 'tai_bias', 'tai_unpool',
                                                                 'tai_convlstm')) and 'set_' not in n and 'floats' not in n and 'transform' not in n
         and 'bytes' not in n and 'variant' not in n]
with torch.no_grad():
    m(5, P, Fo)                                  # warm-up: derived weights, workspaces
    torch.cuda.synchronize()
    for n in names:
        setattr(L, n, wrap(n))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m(5, P, Fo)
    e1.record()
    torch.cuda.synchronize()
total = e0.elapsed_time(e1)
agg = collections.OrderedDict()
for name, ints, a, b in records:
    k = (name, ints)
    t = a.elapsed_time(b)
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += t
inside = sum(v[1] for v in agg.values())
print('eager forward %.2f ms; %d library calls, %.2f ms inside them (event-bracketed: includes launch gaps)' % (total, len(records), inside))
print('%-44s %-44s %5s %9s %8s %7s' % ('entry point', 'integer arguments', 'calls', 'total ms', 'us/call', 'TF'))
for (name, ints), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = ''
    if 'conv3x3_wino' in name and len(ints) >= 5:
        # (N, C, K, H, W, ...) in every 3x3 entry point, possibly behind nparts
        q = ints[1:] if ('parts' in name or '_ex' in name) and ints[0] <= 4 else ints
        N, C, K, H, W = q[0], q[1], q[2], q[3], q[4]
        tf = '%.0f' % (2.0 * N * C * K * 9 * H * W / (t / n * 1e-3) / 1e12)
    print('%-44s %-44s %5d %9.3f %8.1f %7s' % (name, ' '.join(map(str, ints)), n, t, t / n * 1e3, tf))
