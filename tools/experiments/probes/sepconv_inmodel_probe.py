#!/usr/bin/env python3
"""Why is the in-model sepconv launch ([T*B,1,128,128], grid 655360) slower than five times the [32,1,128,128] one?

Times the SAME launch (HIP events on the launch stream, kernels queued back to back so the GPU never idles) behind
different predecessors and with different operand values / batch sizes:

  alone          N back-to-back launches of the sepconv only
  after_conv     the kernel network's last layer (Winograd 51->51 @128^2 over the same batch) then the sepconv
  after_stream   a plain HBM-streaming kernel (copy of 2 GB) then the sepconv
  after_valu     ~1.5 ms of the sepconv itself on other buffers, then the sepconv (VALU + HBM load of the same kind)
  values         random N(0, 0.1^2) taps / zeros / the model's own taps (seeded weights)
  batch          160 in one launch vs 5 launches of 32 vs 10 of 16, each group behind ITS conv (taps just written)

Usage: python tools/sepconv_inmodel_probe.py [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops, synthetic

dev = torch.device('cuda:0')
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
KS, H, W = 51, 128, 128
sep = vfi.SeparableConvolution.apply
torch.backends.cudnn.allow_tf32 = False


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(pre, body, reps=REPS):
    """pre(): kernels queued before the timed body each repetition (not timed); body(): timed.  Returns us per body, list."""
    ts = []
    for _ in range(3):
        pre(); body()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(reps):
        pre()
        a, b = ev(), ev()
        a.record(); body(); b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) * 1e3 for a, b in pairs]
    return float(np.median(ts)), float(np.mean(ts)), float(np.min(ts)), float(np.max(ts))


def fmt(name, r, n32=5.0):
    bytes_ = 220062208.0 * n32
    print('%-44s median %7.1f us  mean %7.1f  min %7.1f  max %7.1f   -> %.3f of 8 TB/s (median)' % (
        name, r[0], r[1], r[2], r[3], bytes_ / (r[0] * 1e-6) / 8e12), flush=True)


def main():
    g = torch.Generator().manual_seed(7)
    N = 160
    inp = (torch.rand(N, 1, H + KS - 1, W + KS - 1, generator=g) * 2 - 1).to(dev)
    v = (torch.randn(N, KS, H, W, generator=g) * 0.1).to(dev)
    h = (torch.randn(N, KS, H, W, generator=g) * 0.1).to(dev)
    v2, h2 = v.clone(), h.clone()
    x51 = torch.randn(N, KS, H, W, generator=g).to(dev) * 0.5
    w51 = (torch.randn(KS, KS, 3, 3, generator=g) / np.sqrt(KS * 9)).to(dev)
    b51 = (torch.randn(KS, generator=g) * 0.01).to(dev)
    big_a = torch.empty(256 << 20, dtype=torch.float32, device=dev)      # 1 GiB
    big_b = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    none = lambda: None
    conv = lambda xs, out=None: conv_ops.conv_bias_act(xs, w51, b51, 1, None, out=out)
    with torch.no_grad():
        print('--- predecessor (launch [160,1,128,128], random taps)')
        fmt('alone (back to back)', timed(none, lambda: sep(inp, v, h, KS)))
        fmt('after 51->51 conv on another buffer', timed(lambda: conv(x51), lambda: sep(inp, v, h, KS)))
        fmt('after conv WRITING v (taps just produced)', timed(lambda: conv(x51, out=v), lambda: sep(inp, v, h, KS)))
        fmt('after conv writing v and conv writing h', timed(lambda: (conv(x51, out=v), conv(x51, out=h)), lambda: sep(inp, v, h, KS)))
        v.copy_(v2); h.copy_(h2)
        fmt('after 2 GiB streaming copy', timed(lambda: big_b.copy_(big_a), lambda: sep(inp, v, h, KS)))
        fmt('after 6 sepconv launches on other taps', timed(lambda: [sep(inp, v2, h2, KS) for _ in range(6)], lambda: sep(inp, v, h, KS)))
        print('--- operand values (alone)')
        vz, hz = torch.zeros_like(v), torch.zeros_like(h)
        fmt('zero taps', timed(none, lambda: sep(inp, vz, hz, KS)))
        fmt('zero taps and zero input', timed(none, lambda: sep(torch.zeros_like(inp), vz, hz, KS)))
        del vz, hz
        # the model's own taps: seeded TAI_gray, B = 32, T = 5
        model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(dev).eval()
        clips = synthetic.make_clips(32, 15, 1, H, W, synthetic.SEEDS['cfg2'])
        P, _, Fo = (torch.from_numpy(a).to(dev) for a in synthetic.split_clip(clips, 5, 5, 5))
        grabbed = []
        orig = model.kernelnet.separableConvolution
        model.kernelnet.separableConvolution = lambda i, vv, hh, ks: (grabbed.append((i.clone(), vv.clone(), hh.clone())), orig(i, vv, hh, ks))[1]
        model(5, P, Fo)
        model.kernelnet.separableConvolution = orig
        mi, mv, mh = grabbed[0]
        print('model taps: v mean %.3g std %.3g  h mean %.3g std %.3g  input std %.3g' % (
            float(mv.mean()), float(mv.std()), float(mh.mean()), float(mh.std()), float(mi.std())))
        fmt('model taps, alone', timed(none, lambda: sep(mi, mv, mh, KS)))
        fmt('model taps, after 51->51 conv (other buffer)', timed(lambda: conv(x51), lambda: sep(mi, mv, mh, KS)))
        del model, grabbed
        print('--- batch split, each group right behind the convs that write ITS taps (timed: sepconv launches only)')
        for nb in (160, 32, 16):
            groups = N // nb
            pairs_all = []
            for rep in range(REPS + 3):
                pairs = []
                for gi in range(groups):
                    s = slice(gi * nb, (gi + 1) * nb)
                    conv(x51[s], out=v[s]); conv(x51[s], out=h[s])
                    a, b = ev(), ev()
                    a.record(); sep(inp[s], v[s], h[s], KS); b.record()
                    pairs.append((a, b))
                pairs_all.append(pairs)
            torch.cuda.synchronize()
            tot = [sum(a.elapsed_time(b) for a, b in pairs) * 1e3 for pairs in pairs_all[3:]]
            r = (float(np.median(tot)), float(np.mean(tot)), float(np.min(tot)), float(np.max(tot)))
            fmt('%3d groups of %3d (sum of sepconv time)' % (groups, nb), r)
            e0, e1 = ev(), ev()
            e0.record()
            for rep in range(REPS):
                for gi in range(groups):
                    s = slice(gi * nb, (gi + 1) * nb)
                    conv(x51[s], out=v[s]); conv(x51[s], out=h[s]); sep(inp[s], v[s], h[s], KS)
            e1.record(); torch.cuda.synchronize()
            print('     whole sequence (2 convs + sepconv per group): %.1f us per 160 samples' % (e0.elapsed_time(e1) * 1e3 / REPS), flush=True)


if __name__ == '__main__':
    main()
