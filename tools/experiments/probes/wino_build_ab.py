"""Same-box A/B of two builds of the Winograd convolution kernel: the in-tree library against build/libtai_orig.so (a build of
another commit: `git stash; hipcc ... -o build/libtai_orig.so ...; git stash pop`), alternating in one process; results must
be bit-identical.  Box-to-box spread is larger than most kernel changes: only this kind of comparison counts."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native
L = _native.lib()
P, I, V = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
orig_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'libtai_orig.so')
O = None
if os.path.exists(orig_path):
    O = ctypes.CDLL(orig_path)
    O.tai_conv3x3_wino_forward.argtypes = [P, P, P, P, I, I, I, I, I, I, V]
    O.tai_conv3x3_wino_forward.restype = I

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3

shapes = [(64, 64, 64, 128, 128), (64, 128, 64, 128, 128), (64, 64, 128, 64, 64), (64, 128, 128, 64, 64), (64, 256, 128, 64, 64),
          (64, 256, 256, 32, 32), (64, 512, 256, 32, 32), (64, 512, 1024, 16, 16), (160, 51, 51, 128, 128), (160, 64, 64, 64, 64), (160, 64, 51, 64, 64)]
for (N, C, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(K, C, 3, 3, generator=g) * 0.05).cuda(); b = torch.randn(K, generator=g).cuda()
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'tw')
    y = torch.empty(N, K, H, W, device='cuda')
    run = lambda: _native.check(L.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s), 'fw')
    run_o = (lambda: O.tai_conv3x3_wino_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s)) if O else None
    out = []
    ref = None
    for reps in (0, 1, 0, 1, 0, 1):
        if reps == 0:
            if run_o is None: continue
            f = run_o
        else:
            f = run
        f(); torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        same = torch.equal(y, ref)
        out.append('%s: %.0f us%s' % ('orig' if reps == 0 else 'new', t(f), '' if same else ' DIFFERS'))
    print('x(%d,%d,%d,%d)->%d  ' % (N, C, H, W, K) + '   '.join(out), flush=True)
