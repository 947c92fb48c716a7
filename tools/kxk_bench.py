"""5x5 / 7x7 MotionEnc layers: displaced reads of a halo plane (round 2) against the stack of shifted copies (round 1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native, conv_ops
L = _native.lib()
dev = 'cuda:0'


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (N, Ci, Co, H, W, k) in ((64, 64, 128, 64, 64, 5), (64, 128, 256, 32, 32, 7)):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    S, top, left, ih, iw = conv_ops.halo_geometry(H, W, k)
    plane = torch.zeros(N, Ci, ih, iw, device=dev)
    plane[:, :, top:top + H, left:left + W] = x
    U = conv_ops._wino_weights_kxk(w)
    y = torch.empty(N, Co, H, W, device=dev)
    yp = torch.empty(N, Co, H // 2, W // 2, device=dev)
    xs = (ctypes.c_void_p * 1)(plane.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    t_disp = timed(lambda: _native.check(L.tai_conv3x3_wino_forward_ex(xs, 1, k, U.data_ptr(), b.data_ptr(), y.data_ptr(), yp.data_ptr(), 0, 0, 0, 0,
                                                                      None, None, N, S * S * Ci, Co, H, W, ih, iw, 1, 2, 1, s), 'ex'))
    ya = y.clone()
    t_stack = timed(lambda: conv_ops._kxk_as_wino(x, w, b, 'relu', True))
    yb, _ = conv_ops._kxk_as_wino(x, w, b, 'relu', True)
    print('%dx%d %d->%d @%dx%d N=%d: displaced reads %.0f us | shift_stack + window conv %.0f us | equal %s' % (k, k, Ci, Co, H, W, N, t_disp, t_stack, bool(torch.equal(ya, yb))))
