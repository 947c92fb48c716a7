#!/usr/bin/env python3
"""Measured on the CPU, before any kernel is written (as tools/split_bf16_study.py was for the split-bf16 arithmetic): what would
Winograd F(4x4, 3x3) in fp32 do to parity?

F(2x2, 3x3) -- csrc/wino_conv.hip.inc -- spends 16 multiplies on a 2 x 2 output tile (2.25x fewer than the direct form's 36);
F(4x4, 3x3) spends 36 on a 4 x 4 tile (4x fewer), i.e. 1.78x fewer MFMAs for the same layer, at the price of transforms with
constants up to 8 (B^T: 4, 5; A^T: 8; G: 1/24) whose fp32 rounding is amplified accordingly.  This tool emulates the bi-TAI
forward's 3x3 / 5x5 / 7x7 convolutions (src/models/mcnet/mcnet.py:28-224, src/models/tai/tai.py:244-348) in

    direct_f32   F.conv2d in fp32
    wino_f32     F(2x2, 3x3), transforms, products and accumulation in fp32      (the arithmetic of csrc/wino_conv.hip.inc)
    wino_f43     F(4x4, 3x3), the same in fp32 (interpolation points 0, +-1, +-2, inf)
    wino_f43big  F(4x4, 3x3) only where it would pay most (C >= 128 and K >= 128), F(2x2, 3x3) elsewhere

and compares each, end to end (full-width TAI_gray, T = 5 and 10, one seeded clip) and per layer, with float64.

Usage: python tools/wino_f43_study.py [--T 5,10] [--threads 8]      (about ten minutes on 8 cores)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import torch
import torch.nn.functional as F

import split_bf16_study as sbs
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from oracle import tai_oracle

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1.]])
G4 = torch.tensor([[1 / 4., 0, 0], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.], [1 / 24., 1 / 12., 1 / 6.],
                   [1 / 24., -1 / 12., 1 / 6.], [0, 0, 1.]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])

_f23 = sbs.wino3x3


def wino3x3_f43(x, w, halo=False):
    """3x3 stride-1 pad-1 convolution as F(4x4, 3x3) in x's dtype; H, W multiples of 4 (halo: x carries its one-pixel border)."""
    N, C, H, W = x.shape
    if halo:
        H, W = H - 2, W - 2
    K = w.shape[0]
    dt = x.dtype
    # the transformed weights are made once per weight (as tai_conv3x3_wino_transform_weights does): in float64, rounded to dt
    U = torch.einsum('ai,kcij,bj->abkc', G4, w.double(), G4).to(dt).reshape(36, K, C)
    bt, at = BT4.to(dt), AT4.to(dt)
    xp = x if halo else F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                         # [N,C,TH,TW,6,6]
    TH, TW = d.shape[2], d.shape[3]
    t = torch.einsum('ai,nctuij->nctuaj', bt, d)                   # column pass, then row pass: two roundings, as a kernel would
    V = torch.einsum('nctuaj,bj->abcntu', t, bt).reshape(36, C, N * TH * TW)
    M = torch.bmm(U, V).reshape(6, 6, K, N, TH, TW)
    t2 = torch.einsum('ia,abkntu->ibkntu', at, M)
    Y = torch.einsum('ibkntu,jb->nktiuj', t2, at)                  # [N,K,TH,4,TW,4]
    return Y.reshape(N, K, H, W)


def wino3x3_dispatch(x, w, mode, halo=False):
    H, W = (x.shape[2] - 2, x.shape[3] - 2) if halo else x.shape[2:]
    if mode in ('f43', 'f43big') and H % 4 == 0 and W % 4 == 0:
        if mode == 'f43' or (w.shape[0] >= 128 and w.shape[1] >= 128):
            return wino3x3_f43(x, w, halo)
    return _f23(x, w, 'f32', halo)


sbs.wino3x3 = wino3x3_dispatch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', default='5,10')
    ap.add_argument('--threads', type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    # self-check of the transforms: exact on small integers
    g = torch.Generator().manual_seed(1)
    xi = torch.randint(-3, 4, (1, 8, 8, 8), generator=g).double(); wi = torch.randint(-2, 3, (4, 8, 3, 3), generator=g).double()
    assert float((wino3x3_f43(xi, wi) - F.conv2d(xi, wi, padding=1)).abs().max()) < 1e-9
    model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd32.items()}
    modes = (('direct_f32', 'direct'), ('wino_f32', 'f32'), ('wino_f43', 'f43'), ('wino_f43big', 'f43big'))
    keys = ('pred', 'pred_forward', 'interp_net_outputs_1')
    print('full-width TAI_gray (gf 64, ks 51, 5 blocks), weights and biases from synthetic.seeded_init(0), one clip of synthetic.make_clips; '
          'every figure: max |x - ref| / max |ref| against the SAME network evaluated in float64', flush=True)
    layer_inputs = None
    for T in [int(t) for t in args.T.split(',')]:
        clips = synthetic.make_clips(1, 5 + T + 5, 1, 128, 128, synthetic.SEEDS['cfg5' if T == 10 else 'cfg2'])
        P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, T, 5))
        t0 = time.time()
        with torch.no_grad(), sbs.Patched('direct', f64=True):
            ref = tai_oracle.tai_forward(sd64, 1, 5, 51, T, P.double(), Fo.double())
        print('\nT = %d: float64 reference in %.0f s' % (T, time.time() - t0), flush=True)
        print('%-12s %s   PSNR(pred, ground truth) dB   uint8 pixels differing from the float64 run' % ('end to end', '  '.join('%-22s' % k for k in keys)))
        ref_u8 = ((ref['pred'].clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
        gt_u8 = ((GT.double().clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8).double()
        psnr = lambda a: 10 * np.log10(255.0 ** 2 / float(((a.double() - gt_u8) ** 2).mean()))
        print('%-12s %s   %.4f' % ('float64', '  '.join('%-22s' % '0' for _ in keys), psnr(ref_u8)))
        for name, mode in modes:
            rec = {} if (layer_inputs is None and mode == 'f32') else None
            t0 = time.time()
            with torch.no_grad(), sbs.Patched(mode, record=rec):
                out = tai_oracle.tai_forward(sd32, 1, 5, 51, T, P, Fo)
            if rec is not None:
                layer_inputs = rec
            errs = [float((out[k].double() - ref[k]).abs().max() / ref[k].abs().max()) for k in keys]
            u8 = ((out['pred'].clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
            print('%-12s %s   %.4f   %d of %d   (%.0f s)' % (name, '  '.join('%-22.3e' % e for e in errs), psnr(u8), int((u8 != ref_u8).sum()), u8.numel(),
                                                          time.time() - t0), flush=True)
    print('\nper layer (the layer\'s own input from the wino_f32 run; reference: the same layer in float64 on that input):')
    print('%-44s %-22s %-12s %-12s %-12s' % ('layer', 'x -> K', 'direct_f32', 'wino_f32', 'wino_f43'))
    worst = {n: 0.0 for n, _ in modes[:3]}
    for key, (x, w, b, pad) in layer_inputs.items():
        K, C, k, _ = w.shape
        if C == 1 or K == 1:
            continue
        with torch.no_grad():
            r = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
            row = []
            for name, mode in modes[:3]:
                e = float((sbs.conv_variant(x, w, b, pad, mode).double() - r).abs().max() / r.abs().max())
                worst[name] = max(worst[name], e)
                row.append(e)
        print('%-44s %-22s %s' % (key, '%s %dx%d -> %d' % (tuple(x.shape), k, k, K), ' '.join('%-12.2e' % e for e in row)), flush=True)
    print('%-44s %-22s %s' % ('worst layer', '', ' '.join('%-12.2e' % worst[n] for n, _ in modes[:3])))


if __name__ == '__main__':
    main()
