"""Same-box A/B of the Winograd kernel's second-output epilogues (EPI 1: y and y2 = y + fixed_unpooling(addx); EPI 2: the sum
alone) between the in-tree library and build/libtai_orig.so (a build of another commit), alternating in one process; results
must be bit-identical.  Shapes: the Residual blocks' last convolutions of configs[1] (mcnet.py:156-185 feeding :234-236)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native
L = _native.lib()
P, I, V = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
O = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'libtai_orig.so'))
for lib in (L, O):
    lib.tai_conv3x3_wino_forward_ex.argtypes = [P, I, I, P, P, P, P, I, I, I, I, P, P] + [I] * 10 + [V]
    lib.tai_conv3x3_wino_forward_ex.restype = I


def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


for (N, C, K, H, W, second) in ((64, 64, 64, 128, 128, False), (64, 128, 128, 64, 64, True), (64, 256, 256, 32, 32, True)):
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(K, C, 3, 3, generator=g) * 0.05).cuda(); b = torch.randn(K, generator=g).cuda()
    addx = torch.randn(N, K, H // 2, W // 2, generator=g).cuda()
    U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'tw')
    y = torch.empty(N, K, H, W, device='cuda'); y2 = torch.empty(N, K, H, W, device='cuda') if second else None
    xs = (ctypes.c_void_p * 1)(x.data_ptr())
    def run(lib):
        rc = lib.tai_conv3x3_wino_forward_ex(xs, 1, 0, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, 0, 0, 0, 0, addx.data_ptr(),
                                             y2.data_ptr() if second else None, N, C, K, H, W, 0, 0, 0, 0, 0, s)
        assert rc == 0, rc
    out, ref = [], None
    for which in (0, 1, 0, 1, 0, 1):
        lib = O if which == 0 else L
        run(lib); torch.cuda.synchronize()
        cur = (y.clone(), y2.clone() if second else None)
        if ref is None: ref = cur
        same = torch.equal(cur[0], ref[0]) and (not second or torch.equal(cur[1], ref[1]))
        out.append('%s: %.0f us%s' % ('orig' if which == 0 else 'new', t(lambda: run(lib)), '' if same else ' DIFFERS'))
    print('x(%d,%d,%d,%d)->%d EPI %d  ' % (N, C, H, W, K, 1 if second else 2) + '   '.join(out), flush=True)
