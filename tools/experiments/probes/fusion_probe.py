import sys, torch, time
import torch.nn.functional as F
dev = 'cuda:0'
shapes = [((64, 64, 128, 128), (64, 64, 3, 3)), ((64, 256, 32, 32), (256, 256, 3, 3)), ((32, 64, 64, 64), (64, 64, 3, 3)),
          ((64, 128, 64, 64), (128, 128, 3, 3)), ((32, 51, 128, 128), (51, 51, 3, 3)), ((64, 512, 16, 16), (1024, 512, 3, 3))]
def tm(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
for xs, ws in shapes:
    x = torch.randn(*xs, device=dev); w = torch.randn(*ws, device=dev) * 0.05; b = torch.randn(ws[0], device=dev)
    t_conv = tm(lambda: F.conv2d(x, w, None, 1, 1))
    t_cb = tm(lambda: F.conv2d(x, w, b, 1, 1))
    t_cbr = tm(lambda: torch.relu_(F.conv2d(x, w, b, 1, 1)))
    try:
        t_fused = tm(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1))
        ref = torch.relu(F.conv2d(x, w, b, 1, 1)); got = torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1)
        err = float((ref - got).abs().max())
    except Exception as e:
        t_fused, err = float('nan'), str(e)[:80]
    xl = x.contiguous(memory_format=torch.channels_last); wl = w.contiguous(memory_format=torch.channels_last)
    t_cl = tm(lambda: torch.relu_(F.conv2d(xl, wl, b, 1, 1)))
    print(xs, ws, 'conv %.3f  +bias %.3f  +bias+relu %.3f  miopen_fused %.3f (err %s)  channels_last+bias+relu %.3f ms' % (t_conv, t_cb, t_cbr, t_fused, err, t_cl), flush=True)
