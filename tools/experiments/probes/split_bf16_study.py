#!/usr/bin/env python3
"""VERDICT r03 item 7 -- measured on the CPU, before any kernel is written: what would a split-bf16 Winograd GEMM do to parity?

bf16 MFMA runs at 16x the fp32 MFMA rate on gfx950, so a product of two fp32 operands emulated by SIX bf16 products
(x = hi + mid + lo, three bf16 terms each; hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid, fp32 accumulation) would be ~2.7x the
fp32 MFMA throughput.  SURVEY.md section 7 allows reduced-precision arithmetic only as an opt-in mode, never on the parity path; the
question here is whether the six-product form is reduced precision at all.  This tool emulates, in PyTorch on the CPU, the 3x3
convolutions of the bi-TAI forward (src/models/mcnet/mcnet.py:28-224, src/models/tai/tai.py:244-348) as Winograd F(2x2, 3x3) GEMMs

    wino_f32      transforms in fp32, products and accumulation in fp32                    (the arithmetic of csrc/wino_conv.hip.inc)
    wino_bf16x3   transformed weights AND patches split into three bf16 terms, six products, fp32 accumulation
    wino_bf16x2   two terms, three products (hi*hi + hi*lo + lo*hi)                       (for context: ~5.3x the fp32 rate)
    direct_f32    F.conv2d in fp32                                                        (the CPU oracle's arithmetic)

and compares each, per layer and end to end (full-width TAI_gray, T = 5 and T = 10, one seeded clip), with the same network evaluated
in float64.  A bf16 x bf16 product is exact in fp32 (8 + 8 mantissa bits), so a fp32 matmul of bf16-rounded operands IS the bf16
MFMA's arithmetic up to the order of the fp32 accumulation.  5x5 / 7x7 layers are cut into 3x3 blocks as the product does; the 1 -> 64
and 64 -> 1 layers stay direct fp32 (they are direct kernels on the GPU).

Usage: python tools/split_bf16_study.py [--T 5,10] [--threads 8]      (about ten minutes on 8 cores)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from oracle import tai_oracle

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])


def split_bf16(x, terms):
    out, rest = [], x
    for _ in range(terms):
        t = rest.to(torch.bfloat16).to(x.dtype)
        out.append(t)
        rest = rest - t
    return out


def wino_gemm(U, V, mode):
    """M[pos] = U[pos] @ V[pos]; U [16,K,C], V [16,C,tiles]."""
    if mode == 'f32':
        return torch.bmm(U, V)
    terms = 3 if mode == 'bf16x3' else 2
    us, vs = split_bf16(U, terms), split_bf16(V, terms)
    if terms == 3:       # smallest products first, all into one fp32 sum
        small = torch.bmm(us[0], vs[2]) + torch.bmm(us[2], vs[0]) + torch.bmm(us[1], vs[1])
        mid = torch.bmm(us[0], vs[1]) + torch.bmm(us[1], vs[0])
        return (small + mid) + torch.bmm(us[0], vs[0])
    return (torch.bmm(us[0], vs[1]) + torch.bmm(us[1], vs[0])) + torch.bmm(us[0], vs[0])


def wino3x3(x, w, mode, halo=False):
    """3x3 stride-1 pad-1 convolution (no bias) as F(2x2, 3x3); x [N,C,H,W] (H, W even), w [K,C,3,3].  halo: x is [N,C,H+2,W+2] and
    carries its own one-pixel border (the shifted copies of the 5x5 / 7x7 layers: true values outside the frame, not zeros)."""
    N, C, H, W = x.shape
    if halo:
        H, W = H - 2, W - 2
    K = w.shape[0]
    dt = x.dtype
    g, bt, at = G.to(dt), BT.to(dt), AT.to(dt)
    U = torch.einsum('ai,kcij,bj->abkc', g, w, g).reshape(16, K, C)
    xp = x if halo else F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # [N,C,TH,TW,4,4]
    TH, TW = d.shape[2], d.shape[3]
    V = torch.einsum('ai,nctuij,bj->abcntu', bt, d, bt).reshape(16, C, N * TH * TW)
    M = wino_gemm(U, V, mode).reshape(4, 4, K, N, TH, TW)
    Y = torch.einsum('ia,abkntu,jb->nktiuj', at, M, at)             # [N,K,TH,2,TW,2]
    return Y.reshape(N, K, H, W)


def conv_variant(x, w, b, pad, mode):
    """k x k stride-1 convolution with padding k // 2 in arithmetic `mode`."""
    K, C, k, _ = w.shape
    if mode == 'direct' or C == 1 or K == 1 or x.shape[2] % 2 or x.shape[3] % 2:
        return F.conv2d(x, w, b, padding=pad)
    if k == 3:
        y = wino3x3(x, w, mode)
    else:               # S x S blocks of 3 x 3 taps, zero past k: block (i, j) sees the input shifted by (3i - p + 1, 3j - p + 1)
        S = (k + 2) // 3
        wp = F.pad(w, (0, 3 * S - k, 0, 3 * S - k))
        H, W = x.shape[2:]
        y = None
        big = F.pad(x, (3 * S + 1, 3 * S + 1, 3 * S + 1, 3 * S + 1))
        o = 3 * S + 1
        for i in range(S):
            for j in range(S):
                dy, dx = 3 * i - pad + 1, 3 * j - pad + 1
                xs = big[:, :, o + dy - 1:o + dy + H + 1, o + dx - 1:o + dx + W + 1]
                t = wino3x3(xs.contiguous(), wp[:, :, 3 * i:3 * i + 3, 3 * j:3 * j + 3].contiguous(), mode, halo=True)
                y = t if y is None else y + t
    return y + b.view(1, -1, 1, 1)


def sepconv_f64(inp_padded, v, h, ks, f64=False):
    """The separable convolution in the tensors' own dtype (float64 reference run): out = sum_fy v[fy] sum_fx h[fx] in[y+fy, x+fx]."""
    B, C, Hp, Wp = inp_padded.shape
    H, W = Hp - ks + 1, Wp - ks + 1
    out = inp_padded.new_zeros(B, C, H, W)
    for fy in range(ks):
        row = inp_padded.new_zeros(B, C, H, W)
        for fx in range(ks):
            row = row + h[:, fx:fx + 1] * inp_padded[:, :, fy:fy + H, fx:fx + W]
        out = out + v[:, fy:fy + 1] * row
    return out


class Patched(object):
    """tai_oracle with its two convolution helpers in arithmetic `mode`; records (key, input) of every layer when asked."""

    def __init__(self, mode, record=None, f64=False):
        self.mode, self.record, self.f64 = mode, record, f64

    def __enter__(self):
        self.saved = (tai_oracle._conv, tai_oracle._convt, tai_oracle.sepconv)
        mode, record = self.mode, self.record

        def _conv(sd, key, x, pad):
            if record is not None and key not in record:
                record[key] = (x.detach().clone(), sd[key + '.weight'], sd[key + '.bias'], pad)
            return conv_variant(x, sd[key + '.weight'], sd[key + '.bias'], pad, mode)

        def _convt(sd, key, x):
            w = sd[key + '.weight'].flip(2, 3).transpose(0, 1).contiguous()        # ConvTranspose2d(k3, s1, p1) == conv with the flipped, transposed weight
            if record is not None and key not in record:
                record[key] = (x.detach().clone(), w, sd[key + '.bias'], 1)
            return conv_variant(x, w, sd[key + '.bias'], 1, mode)
        tai_oracle._conv, tai_oracle._convt = _conv, _convt
        if self.f64:
            tai_oracle.sepconv = sepconv_f64
        return self

    def __exit__(self, *a):
        tai_oracle._conv, tai_oracle._convt, tai_oracle.sepconv = self.saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', default='5,10')
    ap.add_argument('--threads', type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd32.items()}
    modes = (('direct_f32', 'direct'), ('wino_f32', 'f32'), ('wino_bf16x3', 'bf16x3'), ('wino_bf16x2', 'bf16x2'))
    keys = ('pred', 'pred_forward', 'interp_net_outputs_1')
    print('full-width TAI_gray (gf 64, ks 51, 5 blocks), weights and biases from synthetic.seeded_init(0), one clip of synthetic.make_clips; '
          'every figure: max |x - ref| / max |ref| against the SAME network evaluated in float64', flush=True)
    layer_inputs = None
    for T in [int(t) for t in args.T.split(',')]:
        clips = synthetic.make_clips(1, 5 + T + 5, 1, 128, 128, synthetic.SEEDS['cfg5' if T == 10 else 'cfg2'])
        P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, T, 5))
        t0 = time.time()
        with torch.no_grad(), Patched('direct', f64=True):
            ref = tai_oracle.tai_forward(sd64, 1, 5, 51, T, P.double(), Fo.double())
        print('\nT = %d: float64 reference in %.0f s' % (T, time.time() - t0), flush=True)
        print('%-12s %s   PSNR(pred, ground truth) dB   uint8 pixels differing from the float64 run' % ('end to end', '  '.join('%-22s' % k for k in keys)))
        ref_u8 = ((ref['pred'].clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
        mse = lambda a: float(((a.double() - ((GT.double().clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8).double()) ** 2).mean())
        print('%-12s %s   %.4f' % ('float64', '  '.join('%-22s' % '0' for _ in keys), 10 * np.log10(255.0 ** 2 / mse(ref_u8))))
        for name, mode in modes:
            rec = {} if (layer_inputs is None and mode == 'f32') else None
            t0 = time.time()
            with torch.no_grad(), Patched(mode, record=rec):
                out = tai_oracle.tai_forward(sd32, 1, 5, 51, T, P, Fo)
            if rec is not None:
                layer_inputs = rec
            errs = [float((out[k].double() - ref[k]).abs().max() / ref[k].abs().max()) for k in keys]
            u8 = ((out['pred'].clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
            print('%-12s %s   %.4f   %d of %d   (%.0f s)' % (name, '  '.join('%-22.3e' % e for e in errs), 10 * np.log10(255.0 ** 2 / mse(u8)),
                                                          int((u8 != ref_u8).sum()), u8.numel(), time.time() - t0), flush=True)
    # ---- per layer: each distinct 3x3 / 5x5 / 7x7 layer on the input it sees in the fp32 Winograd run of T = first, against float64
    print('\nper layer (the layer\'s own input from the wino_f32 run; reference: the same layer in float64 on that input):')
    print('%-44s %-22s %-12s %-12s %-12s %-12s' % ('layer', 'x -> K', 'direct_f32', 'wino_f32', 'wino_bf16x3', 'wino_bf16x2'))
    worst = {n: 0.0 for n, _ in modes}
    for key, (x, w, b, pad) in layer_inputs.items():
        K, C, k, _ = w.shape
        if C == 1 or K == 1:
            continue
        with torch.no_grad():
            r = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
            row = []
            for name, mode in modes:
                e = float((conv_variant(x, w, b, pad, mode).double() - r).abs().max() / r.abs().max())
                worst[name] = max(worst[name], e)
                row.append(e)
        print('%-44s %-22s %s' % (key, '%s %dx%d -> %d' % (tuple(x.shape), k, k, K), ' '.join('%-12.2e' % e for e in row)), flush=True)
    print('%-44s %-22s %s' % ('worst layer', '', ' '.join('%-12.2e' % worst[n] for n, _ in modes)))


if __name__ == '__main__':
    main()
