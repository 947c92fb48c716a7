#!/usr/bin/env python3
"""us per call of tai_conv3x3_wino_wrw through a library given by PATH (ctypes, no source-hash check): for the ablated builds of the
weight-gradient kernel -- TAI_WRW_ABLATE=noloads|noxform|nobarrier python tools/gen_wino43_asm.py, hipcc -shared -DTAI_ALLOW_ABLATED of csrc/sepconv_capi.hip to a
scratch path, then regenerate without the variable (results of an ablated library are wrong by design; profiles/r05_wrw43_ablation.txt).
Usage: python tools/wrw43_ablate.py path/to/lib.so"""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
I, P, LL = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
lib.tai_conv3x3_wino_wrw_workspace_floats.restype = LL
lib.tai_conv3x3_wino_wrw_workspace_floats.argtypes = [I] * 5
lib.tai_conv3x3_wino_wrw.argtypes = [P, P, P, P, P, I, I, I, I, I, P]
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = []
for (N, C, K, H, W) in [(64, 64, 64, 128, 128), (64, 128, 128, 64, 64), (64, 256, 256, 32, 32), (64, 512, 1024, 16, 16)]:
    x = torch.randn(N, C, H, W).cuda(); go = torch.randn(N, K, H, W).cuda()
    ws = torch.empty(lib.tai_conv3x3_wino_wrw_workspace_floats(N, C, K, H, W), device='cuda')
    dw = torch.empty(K, C, 3, 3, device='cuda'); db = torch.empty(K, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    out.append('%.1f' % timed(lambda: lib.tai_conv3x3_wino_wrw(x.data_ptr(), go.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), N, C, K, H, W, s)))
print(os.path.basename(sys.argv[1]), ' '.join(out), flush=True)
