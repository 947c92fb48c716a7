"""Winograd-MFMA 3x3 convolution (csrc/wino_conv.hip.inc) against MIOpen on the bi-TAI layer shapes: error against an
fp64 convolution, time per call, effective TFLOP/s of direct-convolution flops."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from video_frame_inpainting_amd import _native

L = _native.lib()

def wino(x, w, b, act=1):
    N, C, H, W = x.shape; K = w.shape[0]
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'tw')
    y = torch.empty(N, K, H, W, device='cuda')
    def run():
        _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, act, s), 'fw')
        return y
    return run

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3

shapes = [(2, 8, 8, 8, 8), (3, 20, 51, 12, 20), (1, 65, 64, 16, 16)] if '--quick' in sys.argv else []
shapes += [(64, 64, 64, 128, 128), (64, 128, 64, 128, 128), (64, 128, 128, 64, 64), (64, 256, 128, 64, 64), (64, 256, 256, 32, 32),
           (64, 512, 256, 32, 32), (64, 512, 1024, 16, 16), (32, 51, 51, 128, 128), (32, 64, 64, 64, 64), (32, 64, 51, 64, 64),
           (32, 256, 64, 64, 64), (32, 256, 256, 16, 16), (32, 512, 512, 8, 8), (160, 51, 51, 128, 128), (160, 64, 64, 64, 64)]
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** .5).cuda(); b = torch.randn(K, generator=g).cuda()
    run = wino(x, w, b)
    y = run().clone()
    ref32 = torch.relu(F.conv2d(x, w, b, padding=1))
    if N * C * H * W <= 64 * 64 * 128 * 128:
        ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
        e_mine = (y.double() - ref).abs().max().item(); e_aten = (ref32.double() - ref).abs().max().item()
    else:
        e_mine = (y - ref32).abs().max().item(); e_aten = float('nan')
    tm = t(run); ta = t(lambda: torch.relu_(F.conv2d(x, w, b, padding=1)))
    fl = 2.0 * N * K * C * 9 * H * W
    print('x(%d,%d,%d,%d)->%d  wino %.0f us %.0f TF | miopen %.0f us %.0f TF | err vs f64: wino %.2e miopen %.2e' % (N, C, H, W, K, tm, fl / tm / 1e6, ta, fl / ta / 1e6, e_mine, e_aten), flush=True)

# MotionEnc's 5x5 / 7x7 layers: shift-stack + Winograd 3x3 (conv_ops) against MIOpen's direct form
from video_frame_inpainting_amd import conv_ops
print('k x k layers through the 3x3 Winograd kernel (shifted copies of the input, csrc/thin_conv.hip.inc shift_stack):')
for (N, C, K, H, W, k) in [(64, 64, 128, 64, 64, 5), (64, 128, 256, 32, 32, 7)]:
    g = torch.Generator().manual_seed(k)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(K, C, k, k, generator=g) * (2.0 / (k * k * C)) ** .5).cuda(); b = torch.randn(K, generator=g).cuda()
    with torch.no_grad():
        y = conv_ops.conv_bias_act(x, w, b, k // 2, 'relu')
        ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2))
        r32 = torch.relu(F.conv2d(x, w, b, padding=k // 2))
        tm = t(lambda: conv_ops.conv_bias_act(x, w, b, k // 2, 'relu')); ta = t(lambda: torch.relu_(F.conv2d(x, w, b, padding=k // 2)))
        S = (k + 2) // 3
        st = torch.empty(N, S * S * C, H + 2, W + 4, device='cuda')
        ts = t(lambda: L.tai_conv_shift_stack(x.data_ptr(), st.data_ptr(), N, C, H, W, k, None))
    fl = 2.0 * N * K * C * k * k * H * W
    print('x(%d,%d,%d,%d)->%d %dx%d  shift-stack + wino %.0f us (stack alone %.0f) %.0f TF | miopen %.0f us %.0f TF | err vs f64: wino %.2e miopen %.2e'
          % (N, C, H, W, K, k, k, tm, ts, fl / tm / 1e6, ta, fl / ta / 1e6, (y.double() - ref).abs().max().item(), (r32.double() - ref).abs().max().item()), flush=True)
