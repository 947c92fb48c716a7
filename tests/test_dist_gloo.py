"""world_size-2 data-parallel plumbing on CPU (gloo): gradient all-reduce through flat buckets, replica broadcast
(spectral-norm u vectors included), and that two ranks training on different clips end a step with identical weights."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from video_frame_inpainting_amd import parallel
from video_frame_inpainting_amd.sn_discriminator import SNDiscriminator


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and parallel.world_size() == world
    torch.manual_seed(100 + rank)                     # replicas start DIFFERENT on purpose
    disc = SNDiscriminator((16, 16), 1, 3, 4, 2)
    parallel.materialise_sn_vectors(disc)
    parallel.broadcast_module_state(disc)
    us = [m.u.clone() for m in disc.modules() if hasattr(m, 'Ip')]
    assert all(u is not None for u in us)
    # one "training step" on rank-specific data, gradients averaged through small buckets (forces several buckets)
    opt = torch.optim.Adam(disc.parameters(), lr=1e-3, betas=(0.5, 0.999))
    reducer = parallel.GradAllReducer(disc.parameters(), bucket_bytes=4096)
    assert len(reducer.buckets) > 1
    x = torch.randn(2, 5, 1, 16, 16, generator=torch.Generator().manual_seed(rank))
    loss = disc(x).pow(2).mean()
    loss.backward()
    local_grads = [p.grad.clone() for p in disc.parameters()]
    nbytes = reducer.allreduce_()
    assert nbytes == sum(p.numel() * 4 for p in disc.parameters())
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.numpy() for g in local_grads])
    for i, p in enumerate(disc.parameters()):
        mean = np.mean([gathered[r][i] for r in range(world)], axis=0)
        np.testing.assert_allclose(p.grad.numpy(), mean, rtol=1e-6, atol=1e-7)
    # the overlapped form: gradients accumulate inside the flat buckets, each bucket's all-reduce is launched from the
    # backward pass by the hook of its last gradient; one parameter gets no gradient at all (its bucket still travels).
    # (a plain network: the SN discriminator rescales its weights in every forward, so two passes differ by design)
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Conv2d(1, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    xin = x[:, 0]
    unused = torch.nn.Parameter(torch.ones(3))
    reducer2 = parallel.GradAllReducer(list(net.parameters()) + [unused], bucket_bytes=1024)
    assert len(reducer2.buckets) > 2
    for step in range(2):                                            # the second step reuses the buckets (views in place)
        local = torch.autograd.grad(net(xin).pow(2).mean(), list(net.parameters()))
        gathered2 = [None] * world
        dist.all_gather_object(gathered2, [g.numpy() for g in local])
        reducer2.zero_grad()
        assert all(p.grad is not None and float(p.grad.abs().sum()) == 0 for p in net.parameters())
        launched = []
        orig = reducer2._launch
        reducer2._launch = lambda b, orig=orig: (launched.append(b), orig(b))[1]
        net(xin).pow(2).mean().backward()
        assert len(launched) >= len(reducer2.buckets) - 1            # every bucket but (at most) the one holding `unused`
        reducer2._launch = orig
        assert reducer2.allreduce_() == sum(p.numel() * 4 for p in net.parameters()) + 12
        for i, p in enumerate(net.parameters()):
            mean = np.mean([gathered2[r][i] for r in range(world)], axis=0)
            np.testing.assert_allclose(p.grad.numpy(), mean, rtol=1e-5, atol=1e-7)
        assert float(unused.grad.abs().sum()) == 0
        with torch.no_grad():
            for p in net.parameters():
                p.sub_(0.1 * p.grad)
    # a second backward() before the next zero_grad() would accumulate into buckets whose all-reduce is in flight: refused
    reducer2.zero_grad()
    net(xin).pow(2).mean().backward()
    try:
        net(xin).pow(2).mean().backward()
    except RuntimeError as e:
        assert 'one backward() per zero_grad()' in str(e)
    else:
        raise AssertionError('the second backward() must be refused')
    reducer2.allreduce_()                                            # drains the first backward's collectives on both ranks
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in disc.parameters()] + [u.reshape(-1) for u in us])
    torch.save(flat, os.path.join(out_dir, 'rank%d.pt' % rank))
    assert parallel.allreduce_scalar_mean(float(rank), torch.device('cpu')) == (world - 1) / 2
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / 'rank0.pt')
    b = torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(a, b)            # identical replicas after broadcast + averaged-gradient step


def _env_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend='gloo')
    import video_frame_inpainting_amd as vfi
    from video_frame_inpainting_amd import synthetic
    from video_frame_inpainting_amd.environments import create_training_environment
    torch.manual_seed(7 + rank)                       # different initial weights per rank: sync_replicas must fix that
    np.random.seed(0)                                 # identical (K, T, F) draws
    model = vfi.MCNetFillInModel(4, 1, 3)             # the training environment's G/D step without the GPU-only sepconv
    env = create_training_environment(model, 1, out_dir, 'dp%d' % rank, 3, 2, 3, [32, 32], 1.0, 0.02, 1e-3, 0.5, 4, 2, 3,
                                      [0, 0], device='cpu')
    env.sync_replicas()
    clips = torch.from_numpy(synthetic.make_clips(2, 8, 1, 32, 32, 100 + rank))     # each rank trains on its OWN clips
    for _ in range(2):
        K, T, F = env.sample_KTF(True)
        env.set_train_inputs(clips[:, :K], clips[:, K + T:K + T + F], clips[:, K:K + T])
        env.K, env.T, env.F = K, T, F
        env.train()
        env.forward_train()
        env.optimize_parameters()
    flat = torch.cat([p.detach().reshape(-1) for p in list(env.generator.parameters()) + list(env.discriminator.parameters())])
    torch.save((flat, (K, T, F)), os.path.join(out_dir, 'env_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_training_environment_keeps_replicas_identical(tmp_path):
    """Two ranks, different data and different initial weights: after sync_replicas and two G+D steps with the
    gradient all-reduces of parallel.py, generator and discriminator weights are bit-identical on both ranks."""
    mp.spawn(_env_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, ktf_a = torch.load(tmp_path / 'env_rank0.pt')
    b, ktf_b = torch.load(tmp_path / 'env_rank1.pt')
    assert ktf_a == ktf_b
    assert torch.equal(a, b)


def _shard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend='gloo')
    import video_frame_inpainting_amd as vfi
    from video_frame_inpainting_amd import synthetic
    torch.manual_seed(50 + rank)                      # replicas start different; the broadcast makes them rank 0's
    model = vfi.MCNetFillInModel(4, 1, 3).eval()      # the clip-sharded inference path without the GPU-only sepconv
    parallel.broadcast_module_state(model)
    n_clips = 5                                       # not divisible by 2: the shards differ in size
    clips = torch.from_numpy(synthetic.make_clips(n_clips, 8, 1, 32, 32, 321))
    mine = parallel.shard_slice(n_clips, rank, world)
    with torch.no_grad():
        out = model(2, clips[mine, :3], clips[mine, 5:])['pred']
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine.start, mine.stop, out.numpy()))
    if rank == 0:
        with torch.no_grad():
            whole = model(2, clips[:, :3], clips[:, 5:])['pred'].numpy()
        np.save(os.path.join(out_dir, 'whole.npy'), whole)
        covered = []
        union = np.full_like(whole, np.nan)
        for a, b, o in gathered:
            covered += list(range(a, b))
            union[a:b] = o
        np.save(os.path.join(out_dir, 'union.npy'), union)
        np.save(os.path.join(out_dir, 'covered.npy'), np.array(covered))
    dist.barrier()
    dist.destroy_process_group()


def test_union_of_rank_shards_equals_the_single_rank_forward(tmp_path):
    """Inference shards clips over ranks with no collective on the data path (parallel.shard_slice, predict.py:57): the union of the
    ranks' outputs is the whole batch's forward -- every clip exactly once, values equal (clips are independent: no batch statistic)."""
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    whole, union = np.load(tmp_path / 'whole.npy'), np.load(tmp_path / 'union.npy')
    assert sorted(np.load(tmp_path / 'covered.npy').tolist()) == list(range(5))
    assert np.isfinite(union).all() and float(np.abs(whole).max()) > 0.01
    np.testing.assert_allclose(union, whole, rtol=0, atol=1e-6)
