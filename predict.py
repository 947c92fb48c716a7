#!/usr/bin/env python3
"""Inference driver with the reference's ``predict.py`` surface (reference predict.py:17-120): same flags
(options.TestOptions), same slicing of a clip into preceding / middle / following frames (:49-50), same eval-environment
protocol (:52-56), same PNG outputs and naming -- ``gt_preceding_%04d.png``, ``gt_following_``, ``gt_middle_``,
``pred_middle_`` and, with ``--intermediate_preds``, ``pred_middle_forward_``, ``pred_middle_backward_``,
``interp_net_outputs_{1,2}_`` -- with the truncating uint8 map and the BGR->RGB flip for colour (:103-120).

``--test_video_list_path`` takes the reference's list files (``<path> <a>-<b>``, or ``<path> <a>-<b> <c>-<d>`` with
``--disjoint_clips``; video_frame_inpainting_amd/data.py); a "video" is a directory of frame images or a .npy/.npz frame
array (no video decoder exists in this image).  ``--synthetic N`` (an added flag) runs on N seeded synthetic clips
instead, and ``--random_init`` keeps the seeded xavier weights when no checkpoint exists.  One process per GPU under
``torch.distributed.run`` shards the clips across ranks.

  python predict.py --name demo --K 5 --T 5 --F 5 --c_dim 1 --image_size 128 --batch_size 8 --model_key TAI_gray \
      --qual_result_root results/demo --synthetic 16 --random_init
"""
import os

import numpy as np
import torch
from PIL import Image

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import metrics, parallel, synthetic
from video_frame_inpainting_amd.data import ContiguousVideoClipDataset, DisjointVideoClipDataset
from video_frame_inpainting_amd.environments import create_eval_environment
from video_frame_inpainting_amd.options import TestOptions
from video_frame_inpainting_amd.util import frames_to_uint8


def save_video_frames(video, image_root_dir, image_name_prefix, counter_start=0):
    """video [T,C,H,W] in [-1,1] (BGR if C == 3) -> PNGs (predict.py:103-120)."""
    frames = frames_to_uint8(video)
    os.makedirs(image_root_dir, exist_ok=True)
    for t in range(frames.shape[0]):
        f = frames[t]
        img = Image.fromarray(f[:, :, 0] if f.shape[2] == 1 else f[:, :, ::-1])
        img.save(os.path.join(image_root_dir, '%s_%04d.png' % (image_name_prefix, t + counter_start)))


def main(args=None):
    opt = TestOptions().parse(args, allow_unknown=True)
    vfi.configure_miopen()
    rank, world, local_rank = parallel.init_from_env()
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    if getattr(opt, 'winograd_arithmetic', 'fp32') != 'fp32':
        from video_frame_inpainting_amd import conv_ops
        conv_ops.set_winograd_arithmetic(opt.winograd_arithmetic)
    if getattr(opt, 'winograd_tile', None) in (2, 4):
        from video_frame_inpainting_amd import conv_ops
        conv_ops.set_winograd_tile(opt.winograd_tile)
    H, W = opt.image_size[0] + opt.padding_size[0], opt.image_size[1] + opt.padding_size[1]
    disjoint = bool(getattr(opt, 'disjoint_clips', False)) and not opt.synthetic
    if opt.synthetic:
        total = opt.synthetic
        mine = parallel.shard_slice(total, rank, world)
        clips = torch.from_numpy(synthetic.make_clips(total, opt.K + opt.T + opt.F, opt.c_dim, H, W, opt.seed)[mine])
        labels = ['synthetic_%06d' % i for i in range(total)][mine]
    else:
        if not opt.test_video_list_path:
            raise SystemExit('give --test_video_list_path (reference list format) or --synthetic N')
        if disjoint:                                                 # predict.py:24-29
            dataset = DisjointVideoClipDataset(opt.c_dim, opt.test_video_list_path, opt.K, opt.F, opt.image_size,
                                               opt.padding_size)
        else:
            dataset = ContiguousVideoClipDataset(opt.c_dim, opt.test_video_list_path, opt.K + opt.T + opt.F, False, False,
                                                 opt.image_size, False, opt.padding_size)
        total = len(dataset)
        mine = parallel.shard_slice(total, rank, world)
        items = [dataset[i] for i in range(total)[mine]]
        clips = torch.stack([it['targets'] for it in items]) if items else torch.zeros(0, 1, opt.c_dim, H, W)
        labels = [it['clip_label'] for it in items]
    print('# testing videos = %d (rank %d of %d owns %d)' % (total, rank, world, len(labels)))

    torch.manual_seed(0)
    model = vfi.create_model(opt.model_key)
    env = create_eval_environment(model, opt.checkpoints_dir, opt.name, opt.snapshot_file_name, opt.padding_size,
                                  device=device, load_snapshot=not opt.random_init)
    psnr_rows, ssim_rows = [], []
    h, w = opt.image_size
    for i in range(0, len(labels), opt.batch_size):
        all_frames = clips[i:i + opt.batch_size]
        preceding, following = all_frames[:, :opt.K], all_frames[:, -opt.F:]
        env.set_test_inputs(preceding, following)
        env.T = opt.T
        env.eval()
        env.forward_test()
        out = {k: v.float().cpu() for k, v in env.gen_output.items()}
        gt_middle = None if disjoint else all_frames[:, opt.K:-opt.F]
        if gt_middle is not None:
            p, s, _ = metrics.compute_errors(out['pred'][..., :h, :w].numpy(), gt_middle[..., :h, :w].numpy())
            psnr_rows.append(p)
            ssim_rows.append(s)
        for b in range(all_frames.shape[0]):
            root = os.path.join(opt.qual_result_root, labels[i + b])
            save_video_frames(preceding[b, :, :, :h, :w], root, 'gt_preceding')
            save_video_frames(following[b, :, :, :h, :w], root, 'gt_following', counter_start=opt.K + opt.T)
            if gt_middle is not None:                                # predict.py:67-70: no ground truth for disjoint clips
                save_video_frames(gt_middle[b, :, :, :h, :w], root, 'gt_middle', counter_start=opt.K)
            save_video_frames(out['pred'][b, :, :, :h, :w], root, 'pred_middle', counter_start=opt.K)
            if opt.intermediate_preds:
                for key, prefix in (('pred_forward', 'pred_middle_forward'), ('pred_backward', 'pred_middle_backward'),
                                    ('interp_net_outputs_1', 'interp_net_outputs_1'),
                                    ('interp_net_outputs_2', 'interp_net_outputs_2')):
                    if key in out:
                        save_video_frames(out[key][b, :, :, :h, :w], root, prefix, counter_start=opt.K)
    if psnr_rows:
        pm, pe = metrics.summarize(np.concatenate(psnr_rows))
        sm, se = metrics.summarize(np.concatenate(ssim_rows))
        print('rank %d: PSNR %.4f +- %.4f dB, SSIM %.4f +- %.4f over %d clips' % (rank, pm, pe, sm, se, len(labels)))
    print('Done.')


if __name__ == '__main__':
    main()
