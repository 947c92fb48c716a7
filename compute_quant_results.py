#!/usr/bin/env python3
"""PNG frames written by predict.py -> per-video, per-frame PSNR / SSIM tables (reference compute_quant_results.py:15-61):
same positional arguments (qual_results_root quant_results_root K T [--depth]), same file names read
(``gt_middle_%04d.png`` / ``pred_middle_%04d.png``), same output ``results.npz`` with arrays ``psnr`` [N,T], ``ssim`` [N,T]
and ``video`` [N].  The metrics are the restated scikit-image 0.13.1 definitions of video_frame_inpainting_amd.metrics."""
import argparse
import os

import numpy as np
from PIL import Image

from video_frame_inpainting_amd import metrics


def folder_paths_at_depth(root, depth):
    paths = [root]
    for _ in range(depth):
        paths = [os.path.join(p, d) for p in paths for d in sorted(os.listdir(p)) if os.path.isdir(os.path.join(p, d))]
    return paths


def main(args=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('qual_results_root', type=str)
    parser.add_argument('quant_results_root', type=str)
    parser.add_argument('K', type=int, help='Number of preceding frames')
    parser.add_argument('T', type=int, help='Number of middle frames')
    parser.add_argument('--depth', type=int, default=1,
                        help='Depth of the folders for each video (e.g. 2 for <qual_results_root>/<action>/<video>)')
    args = parser.parse_args(args)
    roots = folder_paths_at_depth(args.qual_results_root, args.depth)
    if len(roots) == 0:
        print('Failed to find any qualitative results (make sure you ran predict.py before this script). Quitting...')
        return
    print('Now computing quantitative results...')
    psnr_table = np.zeros((len(roots), args.T))
    ssim_table = np.zeros((len(roots), args.T))
    for i, root in enumerate(roots):
        for t in range(args.K, args.K + args.T):
            gt_path = os.path.join(root, 'gt_middle_%04d.png' % t)
            if not os.path.exists(gt_path):
                raise RuntimeError('Failed to find GT middle frame at %s (did you generate GT middle frames and use the '
                                   'right values for K and T?)' % gt_path)
            gt = Image.open(gt_path)
            pred = Image.open(os.path.join(root, 'pred_middle_%04d.png' % t))
            psnr_table[i, t - args.K] = metrics.psnr_uint8(np.array(pred), np.array(gt))
            ssim_table[i, t - args.K] = metrics.ssim_uint8(np.array(gt), np.array(pred), multichannel=(gt.mode == 'RGB'))
    os.makedirs(args.quant_results_root, exist_ok=True)
    np.savez(os.path.join(args.quant_results_root, 'results.npz'), psnr=psnr_table, ssim=ssim_table, video=np.array(roots))
    print('Done computing quantitative results.')


if __name__ == '__main__':
    main()
