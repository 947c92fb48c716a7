"""Times the sepconv backward at a given shape through the C ABI with HIP events (hipGraph of back-to-back launches, like
bench.py's forward roofline): all three gradients, the tap gradients alone, and the difference = grad_input; for each
grad_input variant.  Usage: python tools/sepconv_bwd_bench.py [B C H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_frame_inpainting_amd import _native

B, C, H, W = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 1, 128, 128)
ks = 51
L = _native.lib()
dev = 'cuda:0'
g = torch.Generator().manual_seed(7)
inp = (torch.rand(B, C, H + ks - 1, W + ks - 1, generator=g) * 2 - 1).to(dev)
v = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(dev)
h = (torch.randn(B, ks, H, W, generator=g) * 0.1).to(dev)
gO = torch.randn(B, C, H, W, generator=g).to(dev)
gI, gV, gH = torch.empty_like(inp), torch.empty_like(v), torch.empty_like(h)
nbytes = L.tai_sepconv_backward_bytes(B, C, H, W, ks)


def timed(gi, gv, gh, per_graph=20, replays=5):
    def launch():
        _native.check(L.tai_sepconv_backward(gO.data_ptr(), inp.data_ptr(), v.data_ptr(), h.data_ptr(),
                                             gi.data_ptr() if gi is not None else None, gv.data_ptr() if gv is not None else None,
                                             gh.data_ptr() if gh is not None else None, B, C, H, W, ks,
                                             torch.cuda.current_stream().cuda_stream), 'backward')
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(per_graph):
            launch()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        graph.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (per_graph * replays)


for tv, name in ((2, 'patch staged behind a barrier, then the tap loads (round 2)'), (3, 'tap loads at entry, patch by LDS-DMA, gV waves at priority 0 (round 3)'), (0, 'the same with the gV waves at priority 1 (default)'), (4, 'the same with the gV waves at priority 2'), (2, 'again'), (3, 'again'), (0, 'again'), (4, 'again')):
    prev = L.tai_sepconv_set_grad_taps_variant(tv)
    taps = timed(None, gV, gH)
    L.tai_sepconv_set_grad_taps_variant(prev)
    print('[%d,%d,%d,%d] tap gradients (gV + gH), variant %d (%s): %.1f us' % (B, C, H, W, tv, name, taps))
for variant, name in ((0, 'strips, assembly row loop + slab sum (default)'), (4, 'strips, HIP C++ row loop + slab sum'), (2, 'round-1 row scatter + atomics'), (1, 'gather')):
    if variant == 1 and B * C * H * W > 2 ** 19:
        continue
    prev = L.tai_sepconv_set_grad_input_variant(variant)
    allthree = timed(gI, gV, gH)
    L.tai_sepconv_set_grad_input_variant(prev)
    print('grad_input variant %d (%s): all three %.1f us -> gI %.1f us; %.2f TB/s algorithmic = %.1f %% of 8 TB/s on %.0f MB'
          % (variant, name, allthree, allthree - taps, nbytes / allthree / 1e6, nbytes / allthree / 1e6 / 8 * 100, nbytes / 1e6))
