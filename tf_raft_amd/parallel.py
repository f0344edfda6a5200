"""Data-parallel prediction: one process per GPU, image-pair batches sharded across ranks,
predictions all-gathered over RCCL/xGMI (``torch.distributed`` backend ``"nccl"`` IS RCCL on ROCm;
``"gloo"`` is used by the CPU tests).

The reference has no distributed code at all; image pairs are independent end to
end (per-sample instance norm, frozen batch-norm statistics, per-sample correlation volume), so the
only communication is ONE all-gather of the final predictions -- there is no data-path collective
inside the forward pass.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``total`` items for ``rank``; the first ``total % world_size``
    ranks get one extra item (ragged totals are allowed, empty shards too)."""
    if world_size <= 0 or not 0 <= rank < world_size or total < 0:
        raise ValueError(f'bad shard request total={total} rank={rank} world_size={world_size}')
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_batch(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather per-rank shards (dim 0) of a batch of ``total`` items, in rank order.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    sizes = [hi - lo for lo, hi in sizes]
    mx = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f'local shard has {local.shape[0]} items, expected {sizes[dist.get_rank(group)]}')
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


class PendingGather:
    """An all-gather in flight (``all_gather_batch(..., async_op=True)``): the collective runs on the backend's own
    stream next to whatever the caller enqueues afterwards; ``wait()`` makes the current stream wait for it and
    returns the gathered batch.  Holds the send buffer alive until then."""

    def __init__(self, work, out, finish, keep):
        self._work, self._out, self._finish, self._keep = work, out, finish, keep

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
            self._keep = None
            if self._finish is not None:
                self._out = self._finish(self._out)
                self._finish = None
        return self._out


def all_gather_batch_async(local: torch.Tensor, total: int, group=None) -> PendingGather:
    """``all_gather_batch`` without the wait: step i's predictions travel over xGMI while step i + 1 computes."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return PendingGather(None, local, None, None)
    world = dist.get_world_size(group)
    sizes = [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))]
    mx = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f'local shard has {local.shape[0]} items, expected {sizes[dist.get_rank(group)]}')
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=True)
        return PendingGather(work, out, None, local)
    parts = [torch.empty_like(local) for _ in range(world)]
    work = dist.all_gather(parts, local, group=group, async_op=True)
    return PendingGather(work, parts, lambda ps: torch.cat([p[:n] for p, n in zip(ps, sizes)], dim=0), local)


def predict_sharded(predict: Callable[[Sequence], List[torch.Tensor]], image1, image2, group=None,
                    gather_all_iterations: bool = False):
    """Run ``predict([image1_shard, image2_shard])`` on this rank's contiguous shard of the global
    batch and all-gather the result.

    ``image1/2`` are the GLOBAL ``(B, H, W, 3)`` batches (every rank passes the same arrays).
    Returns the gathered ``flow_predictions[-1]`` of shape ``(B, H, W, 2)``, or the list of all
    iterations when ``gather_all_iterations`` is set.
    """
    total = image1.shape[0]
    if total == 0:
        raise ValueError('predict_sharded needs at least one image pair in the global batch (individual ranks may still '
                         'receive an empty shard when the batch is smaller than the world size)')
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(total, rank, world)
    if hi > lo:
        preds = predict([image1[lo:hi], image2[lo:hi]])
    else:
        preds = None
    if world == 1:
        return preds if gather_all_iterations else preds[-1]
    # ranks with an empty shard still have to join the collective with a correctly shaped buffer
    meta = [None]
    if preds is not None:
        meta[0] = (len(preds), tuple(preds[-1].shape[1:]), preds[-1].dtype, str(preds[-1].device))
    metas: List[Optional[tuple]] = [None] * world
    dist.all_gather_object(metas, meta[0], group=group)
    n_iter, tail, dtype, _ = next(m for m in metas if m is not None)
    if preds is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
            else torch.device('cpu')
        preds = [torch.zeros((0,) + tail, dtype=dtype, device=device) for _ in range(n_iter)]
    if gather_all_iterations:
        return [all_gather_batch(torch.as_tensor(p), total, group) for p in preds]
    return all_gather_batch(torch.as_tensor(preds[-1]), total, group)


def all_reduce_gradients(grads, group=None, bucket_bytes: int = 64 << 20):
    """Data-parallel training (BASELINE config 5): average a dict of gradient tensors over the ranks, in place, before
    ``optimizer.apply_gradients``.  The tensors are flattened into buckets of ``bucket_bytes`` (RAFT's 5.26 M parameters =
    21 MB fp32 fit ONE bucket: a single ring all-reduce, per-link bound on xGMI, instead of 154 latency-bound ones), reduced
    with SUM and divided by the world size; iteration order is the sorted parameter names, identical on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    world = dist.get_world_size(group)
    names = sorted(grads)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([grads[n].reshape(-1) for n in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
        off = 0
        for n in bucket:
            k = grads[n].numel()
            grads[n].copy_(flat[off:off + k].view_as(grads[n]))
            off += k
        bucket, size = [], 0

    for n in names:
        nbytes = grads[n].numel() * grads[n].element_size()
        if bucket and size + nbytes > bucket_bytes:
            flush()
        bucket.append(n)
        size += nbytes
    flush()
    return grads


def all_reduce_mean_(tensors, group=None):
    """Average a list of small tensors over the ranks IN PLACE through one flattened all-reduce (the batch-norm batch
    statistics of a data-parallel training step: 15 layers x (mean, variance)).  No-op on one rank."""
    if not tensors or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    return tensors
