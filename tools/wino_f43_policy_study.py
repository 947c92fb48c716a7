#!/usr/bin/env python3
"""Which layers can take Winograd F(4x4, 3x3) without moving the forward's end-to-end error?  Round 4 settled "C >= 128 and K >= 128"
(tools/wino_f43_study.py: wino_f43big); this asks the same question per layer GROUP, on the CPU, before any dispatch changes:
the 5x5 MotionEnc layer (64 -> 128 as 2 x 2 blocks of 3 x 3), the kernel network and the merge residuals (outside MC-Net's recurrence),
MC-Net's layers with 64 output channels.  Every figure: max |x - ref| / max |ref| against the same network in float64.

Usage: python tools/wino_f43_policy_study.py [--T 5,10] [--threads 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'experiments', 'probes'))      # wino_f43_study, split_bf16_study (rounds 4's CPU studies)
import torch

import split_bf16_study as sbs
import wino_f43_study as w43
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from oracle import tai_oracle

KEY = [None]


def big(w):
    return w.shape[0] >= 128 and w.shape[1] >= 128


POLICIES = {
    'f23 everywhere': lambda key, w: False,
    'big (round 4)': lambda key, w: big(w),
    'big + MotionEnc 5x5': lambda key, w: big(w) or key == 'generator.motion_enc.dyn_conv2.1',
    'big + kernelnet + merge': lambda key, w: big(w) or key.startswith(('kernelnet.', 'merge_residual')),
    'big + ME 5x5 + kernelnet + merge': lambda key, w: big(w) or key == 'generator.motion_enc.dyn_conv2.1' or key.startswith(('kernelnet.', 'merge_residual')),
    '... + MC-Net C >= 128': lambda key, w: big(w) or key == 'generator.motion_enc.dyn_conv2.1' or key.startswith(('kernelnet.', 'merge_residual')) or w.shape[1] >= 128,
    '... + MC-Net C >= 64 at <= 64 x 64': None,          # filled below (needs the input's size)
    'f43 everywhere': lambda key, w: True,
}
CURRENT = [None]


def dispatch(x, w, mode, halo=False):
    H, W = (x.shape[2] - 2, x.shape[3] - 2) if halo else x.shape[2:]
    pol = CURRENT[0]
    if pol is not None and H % 4 == 0 and W % 4 == 0:
        if pol == '... + MC-Net C >= 64 at <= 64 x 64':
            take = POLICIES['... + MC-Net C >= 128'](KEY[0], w) or H <= 64
        else:
            take = POLICIES[pol](KEY[0], w)
        if take:
            return w43.wino3x3_f43(x, w, halo)
    return w43._f23(x, w, 'f32', halo)


sbs.wino3x3 = dispatch


class Keyed(sbs.Patched):
    def __enter__(self):
        super().__enter__()
        conv, convt = tai_oracle._conv, tai_oracle._convt

        def _conv(sd, key, x, pad):
            KEY[0] = key
            return conv(sd, key, x, pad)

        def _convt(sd, key, x):
            KEY[0] = key
            return convt(sd, key, x)
        tai_oracle._conv, tai_oracle._convt = _conv, _convt
        return self


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', default='5,10')
    ap.add_argument('--threads', type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    model = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd32.items()}
    keys = ('pred', 'pred_forward', 'pred_backward', 'interp_net_outputs_1')
    for T in [int(t) for t in args.T.split(',')]:
        clips = synthetic.make_clips(1, 5 + T + 5, 1, 128, 128, synthetic.SEEDS['cfg5' if T == 10 else 'cfg2'])
        P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, T, 5))
        CURRENT[0] = None
        with torch.no_grad(), sbs.Patched('direct', f64=True):
            ref = tai_oracle.tai_forward(sd64, 1, 5, 51, T, P.double(), Fo.double())
        print('\nT = %d   %-36s %s' % (T, 'F(4x4, 3x3) on', '  '.join('%-22s' % k for k in keys)), flush=True)
        for pol in POLICIES:
            CURRENT[0] = pol
            t0 = time.time()
            with torch.no_grad(), Keyed('f32'):
                out = tai_oracle.tai_forward(sd32, 1, 5, 51, T, P, Fo)
            errs = [float((out[k].double() - ref[k]).abs().max() / ref[k].abs().max()) for k in keys]
            print('        %-36s %s  (%.0f s)' % (pol, '  '.join('%-22.3e' % e for e in errs), time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
