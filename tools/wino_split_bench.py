"""Split-bf16 Winograd kernel (csrc/wino_split.hip.inc) against the fp32-MFMA kernel on the bi-TAI layer shapes, alternating in one
process: time per call, direct-convolution TFLOP/s, error of both against an fp64 convolution (scaled as tests/test_gpu_wino_conv.py).
python tools/wino_split_bench.py [N,C,K,H,W ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from video_frame_inpainting_amd import _native, conv_ops
L = _native.lib()


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


shapes = [(64, 64, 64, 128, 128), (64, 128, 64, 128, 128), (64, 64, 128, 64, 64), (64, 128, 128, 64, 64), (64, 256, 128, 64, 64),
          (64, 256, 256, 32, 32), (64, 512, 256, 32, 32), (64, 512, 512, 16, 16), (160, 1024, 512, 16, 16), (160, 51, 51, 128, 128),
          (160, 64, 64, 64, 64), (160, 64, 51, 64, 64), (160, 512, 512, 4, 4)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    s = torch.cuda.current_stream().cuda_stream
    Us = {}
    for mode in ('fp32', 'bf16x3'):
        conv_ops.set_winograd_arithmetic(mode)
        U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
        _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'tw')
        Us[mode] = U
    conv_ops.set_winograd_arithmetic('fp32')
    y = torch.empty(N, K, H, W, device='cuda')
    run = {m: (lambda m=m: _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), Us[m].data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s), 'fw'))
           for m in Us}
    small = N * C * H * W <= 2 ** 25
    err = {}
    if small:
        ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
        mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
    times = {m: [] for m in Us}
    for rep in range(3):
        for m in Us:
            run[m](); torch.cuda.synchronize()
            if small and rep == 0:
                err[m] = ((y.double() - ref).abs() / (1 + mag)).max().item()
            times[m].append(t(run[m]))
    fl = 2.0 * N * K * C * 9 * H * W
    line = 'x(%d,%d,%d,%d)->%d ' % (N, C, H, W, K)
    for m in Us:
        best = min(times[m])
        line += '  %s: %7.1f us %6.1f TF/s' % (m, best, fl / best / 1e6) + (' err %.2e' % err[m] if m in err else '')
    line += '   speed-up %.2fx' % (min(times['fp32']) / min(times['bf16x3']))
    print(line, flush=True)
