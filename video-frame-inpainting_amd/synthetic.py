"""Deterministic synthetic clips: there is no network for datasets, so benchmarks and parity tests run
on seeded clips of the dataset's shape.  Smooth low-frequency background + three moving filled rectangles/discs with
constant per-object velocity in [-3, 3] px/frame and intensity in [0.2, 1], + N(0, 0.01) noise, clamped to [-1, 1]."""
import numpy as np

SEEDS = {'cfg1': 1001, 'cfg2': 1002, 'cfg3': 1003, 'cfg4': 1004, 'cfg5': 1005}


def make_clips(B, n_frames, C, H, W, seed):
    """-> float32 [B, n_frames, C, H, W] in [-1, 1]."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    clips = np.empty((B, n_frames, C, H, W), np.float32)
    for b in range(B):
        fy, fx, ph = rs.uniform(0.5, 2.0, 2), rs.uniform(0.5, 2.0, 2), rs.uniform(0, 2 * np.pi, 2)
        bg = 0.25 * np.sin(2 * np.pi * fy[0] * yy / H + ph[0]) * np.cos(2 * np.pi * fx[0] * xx / W) \
            + 0.15 * np.cos(2 * np.pi * fy[1] * yy / H) * np.sin(2 * np.pi * fx[1] * xx / W + ph[1]) - 0.3
        objs = []
        for _ in range(3):
            objs.append(dict(disc=rs.rand() < 0.5, cy=rs.uniform(0.2, 0.8) * H, cx=rs.uniform(0.2, 0.8) * W,
                             vy=rs.uniform(-3, 3), vx=rs.uniform(-3, 3), r=rs.uniform(0.06, 0.16) * min(H, W),
                             col=rs.uniform(0.2, 1.0, C)))
        for t in range(n_frames):
            frame = np.repeat(bg[None], C, axis=0).copy()
            for o in objs:
                cy, cx = o['cy'] + o['vy'] * t, o['cx'] + o['vx'] * t
                if o['disc']:
                    m = (yy - cy) ** 2 + (xx - cx) ** 2 <= o['r'] ** 2
                else:
                    m = (np.abs(yy - cy) <= o['r']) & (np.abs(xx - cx) <= 0.8 * o['r'])
                for c in range(C):
                    frame[c][m] = o['col'][c]
            frame += rs.normal(0, 0.01, frame.shape).astype(np.float32)
            clips[b, t] = np.clip(frame, -1, 1)
    return clips


def split_clip(clips, K, T, F):
    """-> (preceding [:, :K], middle ground truth [:, K:K+T], following [:, K+T:K+T+F])  (predict.py:49-50, train.py:111-114)."""
    return clips[:, :K], clips[:, K:K + T], clips[:, K + T:K + T + F]


def seeded_init(module, seed, bias_std=0.1):
    """Deterministic non-trivial weights AND biases for parity checks and benchmarks (there is no network for the
    published checkpoints): weights ~ N(0, 1/fan_in), biases ~ N(0, bias_std^2), drawn from a private CPU generator in
    sorted parameter-name order, so the result does not depend on the device or on construction order.  The
    reference's own init (util.py:193-196: xavier-normal weights, ZERO biases) makes the kernel network emit taps of
    ~1e-4 and every prediction a constant gray frame -- useless as parity evidence; with these weights the 51 taps per
    pixel are O(0.1) and the predictions span the gray range.  Writes through ``copy_`` under ``no_grad`` so the
    tensors' version counters move (conv_ops caches derived weights per version)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(module.named_parameters()):
            if p.dim() > 1:
                std = 1.0 / np.sqrt(max(p[0].numel(), 1))
            else:
                std = bias_std
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))
    return module
