#!/usr/bin/env python3
"""One-off MIOpen tuning run (GPU box): lets MIOpen's find step benchmark its solvers for every convolution shape of the
bi-TAI forward at the bench batch size, writing the USER find-db under gpurun_out/miopen_tune/db (a small text file that
the package ships as video-frame-inpainting_amd/miopen_db/ and points MIOPEN_USER_DB_PATH at)."""
import os, sys, time
out = os.path.join(os.getcwd(), 'gpurun_out', 'miopen_tune')
os.makedirs(out + '/db', exist_ok=True)
os.environ['MIOPEN_USER_DB_PATH'] = out + '/db'
os.environ.setdefault('MIOPEN_FIND_MODE', 'NORMAL')
import torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
import torch.nn as nn

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t_start = time.time()
torch.manual_seed(0)
m = vfi.create_model('TAI_gray'); m.apply(vfi.util.weights_init); m.to(dev).eval()
clips = synthetic.make_clips(B, 15, 1, 128, 128, 1002)
P, _, Fo = (torch.from_numpy(x).to(dev) for x in synthetic.split_clip(clips, 5, 5, 5))
shapes = {}
orig = F.conv2d
def rec(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    shapes.setdefault((tuple(x.shape), tuple(w.shape), padding if isinstance(padding, int) else tuple(padding)[0]), 0)
    return orig(x, w, b, stride, padding, dilation, groups)
F.conv2d = rec
import video_frame_inpainting_amd.conv_ops as co
co.F.conv2d = rec
with torch.no_grad():
    m(5, P, Fo)
torch.cuda.synchronize()
F.conv2d = orig; co.F.conv2d = orig
print('[tune %.0fs] %d conv shapes' % (time.time() - t_start, len(shapes)), flush=True)

def tm(x, w, p, it=5):
    for _ in range(2): orig(x, w, None, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): orig(x, w, None, 1, p)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it

tot0 = tot1 = 0.0
for i, (xs, ws, p) in enumerate(shapes):
    x = torch.randn(*xs, device=dev); w = torch.randn(*ws, device=dev) * 0.05
    torch.backends.cudnn.benchmark = False
    t0 = tm(x, w, p)
    torch.backends.cudnn.benchmark = True
    t1w = time.time()
    t1 = tm(x, w, p)
    tot0 += t0; tot1 += t1
    print('[tune %.0fs] %2d/%d x%s w%s  immediate %.3f ms -> find %.3f ms  (find took %.0f s)' % (time.time() - t_start, i + 1, len(shapes), xs, ws, t0, t1, time.time() - t1w), flush=True)
    with open(out + '/progress.txt', 'a') as f:
        f.write('%s %s %.4f %.4f\n' % (xs, ws, t0, t1))
print('[tune] sum immediate %.2f ms, sum find %.2f ms' % (tot0, tot1), flush=True)
os.system('ls -la %s/db; du -sh %s/db' % (out, out))
