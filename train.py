#!/usr/bin/env python3
"""Training driver for the bi-TAI path: the step sequence of the reference's ``train.py:102-119`` (sample K,T,F ->
slice the clip -> set_train_inputs -> train() -> forward_train() -> optimize_parameters()), one process per GPU (data
parallel over RCCL when launched with torch.distributed.run), with the reference's snapshot files ``model_latest.ckpt``
/ ``model_%08d.ckpt`` (train.py:137-140).  Clips come from ``--train_video_list_path`` (the reference's list format and
augmentation flags, train.py:37-43; video_frame_inpainting_amd/data.py; each rank shuffles with its own seed) or, with
``--synthetic N``, from N seeded synthetic clips.  TensorBoard logging and the periodic validation of the reference are
outside the hot path.

  python train.py --name demo --K 5 --T 5 --F 5 --c_dim 1 --image_size 128 --batch_size 4 --model_key TAI_gray \
      --max_iter 10 --synthetic 64
"""
import os
import time

import numpy as np
import torch

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import parallel, synthetic
from video_frame_inpainting_amd.data import ContiguousVideoClipDataset
from video_frame_inpainting_amd.environments import create_training_environment
from video_frame_inpainting_amd.options import TrainOptions


def main(args=None):
    opt = TrainOptions().parse(args, allow_unknown=True)
    if getattr(opt, 'miopen_find_mode', None):          # before the first convolution reaches MIOpen
        os.environ['MIOPEN_FIND_MODE'] = opt.miopen_find_mode
    vfi.configure_miopen()                              # FAST find mode unless set; one find-db / kernel cache per rank
    rank, world, local_rank = parallel.init_from_env()
    if getattr(opt, 'winograd_arithmetic', 'fp32') != 'fp32':
        from video_frame_inpainting_amd import conv_ops
        conv_ops.set_winograd_arithmetic(opt.winograd_arithmetic)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    H, W = opt.image_size[0] + opt.padding_size[0], opt.image_size[1] + opt.padding_size[1]
    loader = None
    if getattr(opt, 'train_video_list_path', None) and not opt.synthetic:
        dataset = ContiguousVideoClipDataset(opt.c_dim, opt.train_video_list_path, opt.K + opt.T + opt.F, not opt.no_backwards,
                                             not opt.no_flip, opt.image_size, True, opt.padding_size, seed=opt.seed + 7 * rank)
        gen = torch.Generator().manual_seed(opt.seed + 7 * rank)
        loader = torch.utils.data.DataLoader(dataset, batch_size=opt.batch_size, shuffle=not opt.serial_batches,
                                             num_workers=opt.num_threads, drop_last=True, generator=gen,
                                             worker_init_fn=dataset.worker_init)
        print('# training videos = %d' % len(dataset))

        def batches():                                               # inf_data_loader (train.py:41)
            while True:
                for item in loader:
                    yield item['targets']
        stream = batches()
    else:
        n_clips = opt.synthetic or 64
        clips = torch.from_numpy(synthetic.make_clips(n_clips, opt.K + opt.T + opt.F, opt.c_dim, H, W, opt.seed + rank))

    torch.manual_seed(0)
    np.random.seed(0)          # identical (K, T, F) draws on every rank
    model = vfi.create_model(opt.model_key)
    env = create_training_environment(model, opt.c_dim, opt.checkpoints_dir, opt.name, opt.K, opt.T, opt.F,
                                      opt.image_size, opt.alpha, opt.beta, opt.lr, opt.beta1, opt.df_dim, opt.Ip,
                                      opt.disc_window_size, opt.padding_size, device=device,
                                      graph_step=opt.graph_step)
    env.sync_replicas()
    total_updates = env.start_update
    order = np.random.RandomState(opt.seed + 7 * rank)
    while total_updates < opt.max_iter:
        t0 = time.time()
        total_updates += 1
        env.total_updates = total_updates
        K, T, F = env.sample_KTF(opt.sample_KTF)
        all_frames = next(stream) if loader is not None else clips[order.randint(0, n_clips, opt.batch_size)]
        env.K, env.T, env.F = K, T, F
        env.train()
        # set_train_inputs -> forward_train -> optimize_parameters; one hipGraph replay per update with --graph_step
        env.train_step(all_frames[:, :K], all_frames[:, K + T:K + T + F], all_frames[:, K:K + T])
        if total_updates % opt.print_freq == 0 or total_updates == 1:
            torch.cuda.synchronize()
            errs = env.get_current_errors()
            if rank == 0:
                print('iter %d (K,T,F)=(%d,%d,%d) %.3fs  %s' % (total_updates, K, T, F, time.time() - t0,
                                                                ' '.join('%s=%.5f' % kv for kv in sorted(errs.items()))))
        if total_updates % opt.save_latest_freq == 0:
            env.save('model_latest.ckpt', total_updates, 0, 0)
            env.save('model_%08d.ckpt' % total_updates, total_updates, 0, 0)
    env.save('model_latest.ckpt', total_updates, 0, 0)
    print('Done.')


if __name__ == '__main__':
    main()
