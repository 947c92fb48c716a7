"""PSNR / SSIM as the reference measures them (train.py:256-285; compute_quant_results.py:38-50): on uint8 frames
obtained by a TRUNCATING cast of 255 * (clip(x, -1, 1) + 1) / 2, with scikit-image 0.13.1's ``compare_psnr`` and
``compare_ssim`` defaults restated here (skimage is not a dependency of this package):
  PSNR = 10 log10(255^2 / MSE);
  SSIM: 7x7 uniform window, K1 = 0.01, K2 = 0.03, L = 255, sample (N/(N-1)) covariance, mean over the interior that
  excludes a (win-1)/2 border, channel mean when multichannel.
Host-side float64 numpy, exactly as in the reference (skimage works in float64 on the CPU).
"""
import numpy as np


def to_uint8(x):
    """x in [-1, 1] (any shape) -> uint8 with the reference's truncation (train.py:279-280)."""
    return ((np.clip(np.asarray(x, dtype=np.float32), -1, 1) + 1.) / 2 * 255).astype('uint8')


def psnr_uint8(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return float('inf') if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def _uniform_filter_valid(x, win):
    """Mean over win x win windows, 'valid' region only ([H-win+1, W-win+1]) via an integral image."""
    H, W = x.shape
    c = np.zeros((H + 1, W + 1), dtype=np.float64)
    c[1:, 1:] = np.cumsum(np.cumsum(x, axis=0), axis=1)
    s = c[win:, win:] - c[:-win, win:] - c[win:, :-win] + c[:-win, :-win]
    return s / (win * win)


def ssim_uint8(a, b, multichannel=False, win=7):
    if multichannel:
        return float(np.mean([ssim_uint8(a[..., c], b[..., c], False, win) for c in range(a.shape[-1])]))
    X = a.astype(np.float64)
    Y = b.astype(np.float64)
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    ux, uy = _uniform_filter_valid(X, win), _uniform_filter_valid(Y, win)
    uxx, uyy, uxy = _uniform_filter_valid(X * X, win), _uniform_filter_valid(Y * Y, win), _uniform_filter_valid(X * Y, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    return float(S.mean())


def compute_errors(pred, gt):
    """pred, gt: [B, T, C, H, W] arrays in [-1, 1].  Returns (psnr[B,T], ssim[B,T], l2[B,T]) as train.py:237-287."""
    pred = np.clip(np.asarray(pred, dtype=np.float32), -1, 1)
    gt = np.clip(np.asarray(gt, dtype=np.float32), -1, 1)
    B, T, C = pred.shape[:3]
    psnr, ssim, l2 = np.zeros((B, T)), np.zeros((B, T)), np.zeros((B, T))
    for b in range(B):
        for t in range(T):
            p = np.transpose(pred[b, t], (1, 2, 0))
            g = np.transpose(gt[b, t], (1, 2, 0))
            pf, gf = (p + 1.) / 2, (g + 1.) / 2
            l2[b, t] = ((pf - gf) ** 2).mean()
            pu, gu = (pf * 255).astype('uint8'), (gf * 255).astype('uint8')
            if C == 1:
                pu, gu = pu[..., 0], gu[..., 0]
            psnr[b, t] = psnr_uint8(pu, gu)
            ssim[b, t] = ssim_uint8(gu, pu, multichannel=C > 1)
    return psnr, ssim, l2


def summarize(table):
    """Mean over frames, then mean +- std/sqrt(N) over videos (summarize_quant_results.py:223-228)."""
    per_video = np.mean(table, axis=1)
    return float(np.mean(per_video)), float(np.std(per_video) / np.sqrt(len(per_video)))
