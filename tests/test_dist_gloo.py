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
