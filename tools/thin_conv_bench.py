"""Times the thin-layer direct convolutions against MIOpen's on the configs[1] shapes (64 clips = 2 directions x 32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from video_frame_inpainting_amd.conv_ops import conv_bias_act

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3

for N in (64, 160):
    for (ci, co, k, act) in ((1, 64, 5, 'relu'), (1, 64, 3, 'relu'), (64, 1, 3, 'tanh')):
        x = torch.randn(N, ci, 128, 128, device='cuda'); w = torch.randn(co, ci, k, k, device='cuda') * .1; b = torch.randn(co, device='cuda')
        with torch.no_grad():
            mine = t(lambda: conv_bias_act(x, w, b, k // 2, act))
            def ref():
                y = F.conv2d(x, w, b, padding=k // 2)
                return torch.relu_(y) if act == 'relu' else torch.tanh_(y)
            aten = t(ref)
        wide = N * 64 * 128 * 128 * 4
        print('N=%d %d->%d %dx%d  direct %.1f us (%.2f TB/s of the wide stream)   MIOpen+ATen %.1f us' % (N, ci, co, k, k, mine, wide / mine / 1e6, aten), flush=True)
