"""The collectives the data-parallel path issues, executed on the RCCL backend (torch.distributed "nccl") -- with the one rank a
one-GPU box allows.  A world of one makes every collective an identity, so this is not a test of the arithmetic (the gloo
world-2 tests are); it is the first execution of the backend the 8-GPU runs use: process-group creation over RCCL on this
image, the asynchronous all-reduce of a flat gradient bucket from inside a backward hook with its wait on the compute stream,
broadcast of parameters and of a non-persistent buffer, barrier and the MAX-reduction of bench.py's timing."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%r, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    from video_frame_inpainting_amd import parallel
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    dev = torch.device('cuda:0')
    # bench.py's fence and timing reduction
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    # the reducer's launch: a flat bucket, gradients are views of it, the all-reduce goes out from the last gradient's hook
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(1, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1)).to(dev)
    red = parallel.GradAllReducer(net.parameters(), bucket_bytes=256)
    assert len(red.buckets) > 1
    x = torch.randn(2, 1, 16, 16, device=dev)
    want = torch.autograd.grad(net(x).pow(2).mean(), list(net.parameters()))
    red.zero_grad()
    launched = []
    real_world = parallel.world_size
    parallel.world_size = lambda: 2                    # take the multi-rank code path; the backend still has one rank
    orig = red._launch
    red._launch = lambda b: (launched.append(b), orig(b))[1]
    net(x).pow(2).mean().backward()
    assert len(launched) == len(red.buckets), (launched, len(red.buckets))      # every bucket left from a hook, during backward
    nbytes = red.allreduce_()                          # waits on the compute stream, divides by the (pretended) world of 2
    parallel.world_size = real_world
    torch.cuda.synchronize()
    assert nbytes == sum(p.numel() * 4 for p in net.parameters())
    for p, g in zip(net.parameters(), want):
        assert torch.allclose(p.grad, g / 2, rtol=1e-6, atol=1e-8)              # sum over one rank, divided by two
    # replica broadcast: parameters and a non-persistent buffer (the spectral-norm u vectors travel this way)
    m = torch.nn.Linear(4, 3).to(dev)
    m.register_buffer('u', torch.randn(1, 3, device=dev), persistent=False)
    before = [t.clone() for t in list(m.parameters()) + [m.u]]
    for tns in list(m.parameters()) + [m.u]:
        dist.broadcast(tns.detach(), src=0)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, list(m.parameters()) + [m.u]))
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL single-rank ok')
''')


def test_collectives_run_on_the_rccl_backend(tmp_path):
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / 'rccl_one_rank.py'
    script.write_text(SCRIPT % (ROOT, str(port)))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and 'RCCL single-rank ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
