"""Per-chunk clocks of the Winograd weight-gradient kernel from its shader-clock stamps (wave 0 of every workgroup).
Tools build only: python tools/wrw_timeline.py  [N,C,K,H,W ...]"""
import ctypes, os, sys
os.environ['TAI_NATIVE_TIMING_LIB'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_frame_inpainting_amd import _native
L = _native.lib()
P, I = ctypes.c_void_p, ctypes.c_int
L.tai_conv3x3_wino_wrw_timeline.argtypes = [P, P, P, P, I, I, I, I, I, P, P]
L.tai_conv3x3_wino_wrw_timeline.restype = I
shapes = [(32, 64, 64, 128, 128), (32, 256, 256, 32, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for N, C, K, H, W in shapes:
    x = torch.randn(N, C, H, W, device='cuda'); go = torch.randn(N, K, H, W, device='cuda')
    ws = torch.empty(L.tai_conv3x3_wino_wrw_workspace_floats(N, C, K, H, W), device='cuda')
    dw = torch.empty(K, C, 3, 3, device='cuda')
    kb, cb = (K + 63) // 64, (C + 63) // 64
    nchunks = N * (H // 2) * (W // 2) // 8
    want = max(1, min(256 // (kb * cb), nchunks))
    cps = (nchunks + want - 1) // want
    wgs = kb * cb * ((nchunks + cps - 1) // cps)
    st = torch.zeros(wgs * 64, dtype=torch.int64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _native.check(L.tai_conv3x3_wino_wrw_timeline(x.data_ptr(), go.data_ptr(), dw.data_ptr(), ws.data_ptr(), N, C, K, H, W, st.data_ptr(), s), 'tl')
    torch.cuda.synchronize()
    t = st.cpu().numpy().reshape(wgs, 64).astype(np.float64)
    n = min(cps, 26)
    ch = np.diff(np.concatenate([t[:, 1:2], t[:, 4:4 + n]], axis=1), axis=1)
    print('N%d C%d K%d %dx%d: %d workgroups x %d chunks: prologue %.0f loop %.0f (%.0f/chunk) epilogue %.0f | per-chunk medians: %s'
          % (N, C, K, H, W, wgs, cps, np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 2] - t[:, 1]) / cps,
             np.median(t[:, 3] - t[:, 2]), ' '.join('%.0f' % v for v in np.median(ch, axis=0))), flush=True)
    for c in range(2):
        gs = np.diff(np.concatenate([t[:, 4 + 1 + c:4 + 2 + c], t[:, 30 + 16 * c:46 + 16 * c]], axis=1), axis=1)
        print('   chunk %d groups: %s' % (2 + c, ' '.join('%.0f' % v for v in np.median(gs, axis=0))))
