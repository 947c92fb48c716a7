"""Data parallelism for the bi-TAI path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for the tests).  The reference is single-GPU (no collective anywhere), so this is new
capability shaped by the path itself:

  * inference shards CLIPS: every rank runs the same weights on its own slice of the batch; there is no collective on
    the data path (clips are independent: no BatchNorm, nothing crosses the batch dimension);
  * training adds exactly two gradient all-reduces per step, in the reference's G-then-D order (environments.py:348-355):
    generator grads (38.3 M fp32 = 153 MB for TAI_gray) after ``loss_G.backward()`` and discriminator grads (2.8 M fp32)
    after ``loss_D.backward()``.  Gradients are packed into a few LARGE flat buckets (default 64 MB): xGMI is
    point-to-point, a ring all-reduce is per-link bound (2*(N-1)/N * bytes / ~153 GB/s), so fewer, larger messages
    amortise the per-collective latency; at 153 MB the ring costs ~1.9 ms against a multi-hundred-ms fp32 conv step;
  * replicas must stay bit-identical: weights, the spectral-norm ``u`` vectors (random on first use in the reference)
    and the sampled (K, T, F) are broadcast / derived from rank 0.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, local_rank=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run).
    Returns (rank, world_size, local_rank); a single un-launched process gets (0, 1, 0) and no group.  ``backend`` and
    ``local_rank`` override what the environment implies (`python bench.py --gpus 2 --rehearse-one-gpu`: gloo, every rank on cuda:0)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_slice(n_items, rank_, world):
    """Contiguous slice of ``n_items`` clips owned by ``rank_`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank_ * base + min(rank_, rem)
    return slice(start, start + base + (1 if rank_ < rem else 0))


def broadcast_module_state(module, src=0):
    """Make every replica's parameters AND buffers (persistent or not, e.g. the SN ``u`` vectors) equal to rank src's."""
    if world_size() == 1:
        return
    from . import conv_ops
    for t in list(module.parameters()) + [b for b in module.buffers() if b is not None]:
        dist.broadcast(t.detach(), src=src)        # detach() shares the version counter with t; .data would not
    conv_ops.invalidate_derived(module)            # belt and braces: derived weights are rebuilt from the received ones


def materialise_sn_vectors(discriminator):
    """Draw every spectral-norm ``u`` now (the reference draws it lazily inside the first forward), so that it can be
    broadcast before the first step."""
    for m in discriminator.modules():
        if hasattr(m, 'Ip') and hasattr(m, 'u') and m.u is None:
            m.u = torch.randn(1, m.weight.size(0), device=m.weight.device, dtype=m.weight.dtype)


class GradAllReducer(object):
    """Averages the gradients of ``params`` across ranks through flat buckets of at most ``bucket_bytes``.

    Overlapped form (what the training environments use): ``zero_grad()`` before the backward pass makes every
    ``p.grad`` a VIEW into its bucket's flat buffer (zeroed in one memset per bucket), so autograd accumulates straight
    into the bucket and nothing is gathered or scattered afterwards; a post-accumulate hook per parameter counts a
    bucket's gradients in and launches its asynchronous all-reduce the moment the last one has landed -- buckets are
    filled in reverse parameter order, the order backward produces them, so the first collective starts while most of
    the backward pass is still running.  ``allreduce_()`` after ``backward()`` launches what is left (buckets holding a
    parameter that received no gradient this step), waits, and divides by the world size.

    xGMI is point-to-point: a ring all-reduce is per-link bound (2 (N-1)/N x bytes / ~153 GB/s), so the buckets are few
    and large (default 64 MB: three for the 153 MB of generator gradients).

    Plain form: without a preceding ``zero_grad()``, ``allreduce_()`` flattens whatever gradients exist, reduces and
    copies back (single call, no hooks)."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):                      # backward produces gradients roughly last-layer first
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self.timing = None           # a list: allreduce_() appends (start, end) device events (bench.py: exposed communication)
        self._flat = None            # per bucket: the flat gradient buffer
        self._pending = None         # per bucket: gradients still to arrive this step
        self._work = None            # per bucket: the async all-reduce handle, once launched
        self._armed = False
        self._bucket_of = {}
        self._hooks = []

    # ---- overlapped form
    def _materialise(self):
        self._flat = []
        for b, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            flat = torch.zeros(n, dtype=bucket[0].dtype, device=bucket[0].device)
            off = 0
            for p in bucket:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._bucket_of[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self._flat.append(flat)

    def zero_grad(self):
        """Call instead of ``optimizer.zero_grad()``: zero gradients that live in the buckets, hooks armed."""
        if self._flat is None or any(p.grad is None or p.grad.data_ptr() != f_ptr for p, f_ptr in self._view_ptrs()):
            for h in self._hooks:
                h.remove()
            self._hooks, self._bucket_of = [], {}
            self._materialise()
        else:
            for flat in self._flat:
                flat.zero_()
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._armed = True

    def _view_ptrs(self):
        for flat, bucket in zip(self._flat, self.buckets):
            off = 0
            for p in bucket:
                yield p, flat.data_ptr() + off * flat.element_size()
                off += p.numel()

    def _launch(self, b):
        if world_size() > 1 and self._work[b] is None:
            self._work[b] = dist.all_reduce(self._flat[b], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self._bucket_of[p]
        if world_size() > 1 and (self._pending[b] <= 0 or self._work[b] is not None):
            # a second backward() since zero_grad(): this gradient was just accumulated into a bucket whose all-reduce is
            # already in flight (or complete) -- the collective's result would be undefined
            raise RuntimeError('GradAllReducer: a gradient arrived for a bucket whose all-reduce was already launched; the '
                               'overlapped form supports exactly one backward() per zero_grad()')
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    # ---- both forms
    def allreduce_(self):
        """Launch what has not been launched, wait, divide by the world size; returns the bytes reduced.  With ``self.timing`` set to a
        list, a pair of device events brackets the call on the current stream: backward's kernels are already queued in front of
        the first, the optimizer's behind the second, so their distance is the time the compute stream stood waiting for the
        collectives (plus the divide kernels) -- the communication the backward pass did not hide."""
        ev = None
        if self.timing is not None and self.params and self.params[0].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        total = self._allreduce()
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        return total

    def _allreduce(self):
        world = world_size()
        if self._armed:
            self._armed = False
            total = sum(f.numel() * f.element_size() for f in self._flat)
            if world == 1:
                return 0
            for b in range(len(self.buckets)):               # buckets with a parameter that got no gradient: zeros travel
                self._launch(b)
            for b, work in enumerate(self._work):
                work.wait()
                self._flat[b].div_(world)
            return total
        if world == 1:
            return 0
        total = 0
        handles = []
        for bucket in self.buckets:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
            total += flat.numel() * flat.element_size()
        for work, flat, bucket in handles:
            work.wait()
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        return total


def allreduce_scalar_mean(value, device):
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item() / world_size())
