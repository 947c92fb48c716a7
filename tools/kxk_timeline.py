"""Per-chunk clocks of the displaced-read (PARTS = 2) Winograd kernel against the same layer over a stack of shifted copies
(window mode), from the kernel's shader-clock stamps.  Tools build only: TAI_NATIVE_TIMING_LIB=1 python tools/kxk_timeline.py"""
import ctypes, os, sys
os.environ['TAI_NATIVE_TIMING_LIB'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_frame_inpainting_amd import _native, conv_ops
L = _native.lib()
L.tai_conv3x3_wino_ex_timeline_target.argtypes = [ctypes.c_void_p]
dev = 'cuda:0'
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
    s = torch.cuda.current_stream().cuda_stream
    wgs = (N * H * W // 4 // 32) * (Co // 128)
    nch = S * S * Ci // 8
    xs = (ctypes.c_void_p * 1)(plane.data_ptr())
    for mode in ('displaced', 'stack'):
        st = torch.zeros(wgs * 64 * 2, dtype=torch.int64, device=dev)
        for _ in range(3):
            if mode == 'displaced':
                L.tai_conv3x3_wino_ex_timeline_target(st.data_ptr())
                _native.check(L.tai_conv3x3_wino_forward_ex(xs, 1, k, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, 0, 0, 0, 0, None, None,
                                                           N, S * S * Ci, Co, H, W, ih, iw, 1, 2, 1, s), 'ex')
            else:
                stack = torch.empty((N, S * S * Ci, H + 2, W + 4), device=dev)
                _native.check(L.tai_conv_shift_stack(x.data_ptr(), stack.data_ptr(), N, Ci, H, W, k, s), 'stack')
                # timeline entry: plain tensor, so time the window conv through its stamps variant on the stack as an ordinary input
                _native.check(L.tai_conv3x3_wino_forward_timeline(stack[:, :, 1:H + 1, 2:W + 2].contiguous().data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(),
                                                                 N, S * S * Ci, Co, H, W, st.data_ptr(), s), 'tl')
        torch.cuda.synchronize()
        t = st.cpu().numpy()[:wgs * 64].reshape(wgs, 64).astype(np.float64)
        n = min(nch, 26)
        ch = np.diff(np.concatenate([t[:, 1:2], t[:, 4:4 + n]], axis=1), axis=1)
        print('%dx%d %s: prologue %.0f loop %.0f (%d chunks, %.0f/chunk) epilogue %.0f | per-chunk medians: %s'
              % (k, k, mode, np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), nch, np.median(t[:, 2] - t[:, 1]) / nch, np.median(t[:, 3] - t[:, 2]),
                 ' '.join('%.0f' % v for v in np.median(ch, axis=0))))
