"""Same-box A/B of two builds of the Winograd kernel on channel-PART inputs (the operands of a torch.cat: Residual, CombLayers, the
kernel network's 1024-channel input): the in-tree library against build/libtai_orig.so (a build of another commit), alternating in
one process; results must be bit-identical.  python tools/wino_parts_ab.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native
L = _native.lib()
P, I, V = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
orig_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'libtai_orig.so')
O = ctypes.CDLL(orig_path)
O.tai_conv3x3_wino_forward_parts.argtypes = [P, I, P, P, P, I, I, I, I, I, I, V]
O.tai_conv3x3_wino_forward_parts.restype = I


def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


# (N, parts, channels per part, K, H, W)
shapes = [(64, 2, 64, 64, 128, 128), (64, 2, 128, 128, 64, 64), (64, 2, 256, 256, 32, 32), (64, 2, 64, 128, 64, 64), (160, 4, 256, 512, 16, 16),
          (64, 2, 512, 512, 16, 16)]
for (N, np_, cp, K, H, W) in shapes:
    g = torch.Generator().manual_seed(N + cp)
    C = np_ * cp
    parts = [torch.randn(N, cp, H, W, generator=g).cuda() for _ in range(np_)]
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.05).cuda(); b = torch.randn(K, generator=g).cuda()
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'tw')
    y = torch.empty(N, K, H, W, device='cuda')
    ptrs = (ctypes.c_void_p * 4)(*[p.data_ptr() for p in parts] + [None] * (4 - np_))
    run = lambda: _native.check(L.tai_conv3x3_wino_forward_parts(ptrs, np_, U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s), 'fw')
    run_o = lambda: O.tai_conv3x3_wino_forward_parts(ptrs, np_, U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, 1, s)
    out, ref = [], None
    for which in (0, 1, 0, 1, 0, 1):
        f = run_o if which == 0 else run
        f(); torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        out.append('%s: %.0f us%s' % ('orig' if which == 0 else 'new', t(f), '' if torch.equal(y, ref) else ' DIFFERS'))
    print('%d x %d parts of %d @%dx%d -> %d  ' % (N, np_, cp, H, W, K) + '   '.join(out), flush=True)
