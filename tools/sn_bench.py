#!/usr/bin/env python3
"""Time the spectral-norm renormalisation of the discriminator's five layers: the reference's matmul form as ATen ops vs
tai_sn_power_iteration (wall time per call, launches included)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import sn_discriminator as sn

dev = torch.device('cuda:0')
shapes = [(64, 48), (128, 1024), (256, 2048), (512, 4096), (1, 32768)]
ips = [3, 3, 3, 3, 1]


def ref(W, u, Ip):
    _u = u
    for _ in range(Ip):
        _v = torch.matmul(_u, W); _v = _v / ((_v ** 2).sum() ** 0.5 + 1e-12)
        _u = torch.matmul(_v, W.t()); _u = _u / ((_u ** 2).sum() ** 0.5 + 1e-12)
    return torch.matmul(torch.matmul(_v, W.t()), _u.t()), _u


def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6


for (o, i), Ip in zip(shapes, ips):
    W = torch.randn(o, i, device=dev) * 0.05
    u = torch.randn(1, o, device=dev)
    s0, u0 = ref(W, u, Ip)
    layer = sn.SNLinear(i, o, Ip=Ip).to(dev)
    with torch.no_grad():
        layer.weight.copy_(W)
    layer.u = u.clone()
    with torch.no_grad():
        layer._renormalise_()
    s1, u1 = layer.last_sigma.clone(), layer.u.clone()

    def product():
        with torch.no_grad():
            layer._renormalise_()
    print('W %4dx%5d  ATen matmul form %7.1f us   tai_sn_power_iteration %7.1f us   sigma %.7f %.7f  du %.2e' % (
        o, i, timeit(lambda: ref(W, u, Ip)), timeit(product), float(s0), float(s1), float((u0 - u1).abs().max())), flush=True)
