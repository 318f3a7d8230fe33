"""Impression-parallel data parallelism over RCCL/xGMI (one process per GPU; backend "nccl" is RCCL
on ROCm, "gloo" on CPU for tests).

The reference has no distributed code at all; the path shards naturally by
impression/user: weights (0.66-3.15 M parameters) are replicated, every rank encodes and scores its own
impressions, and FORWARD/INFERENCE NEEDS NO COLLECTIVE.  The grad step (xnrs/training.py:402-431) adds
exactly two:
  1. a differentiable all-gather of the user embeddings, so the in-batch InfoNCE
     (training.py:433-472) sees the GLOBAL batch and equals the single-GPU loss;
  2. one flat fp32 all-reduce (SUM) of the gradients (2.6-12.6 MB: latency-bound -> a single bucket).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous block of impressions owned by `rank` (remainder spread over the first ranks)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, rank: int, world: int):
    """Slice every per-impression entry of a reference-style batch dict (dataset.py:67-158)."""
    def n_of(b):
        return b["targets"].shape[0]

    lo, hi = shard_range(n_of(batch), rank, world)

    def cut(v):
        if isinstance(v, torch.Tensor):
            return v[lo:hi]
        if isinstance(v, dict):
            return {k: cut(x) for k, x in v.items()}
        if isinstance(v, tuple):
            return tuple(cut(x) for x in v)
        if isinstance(v, list):
            return v[lo:hi]
        return v

    return cut(batch)


class _AllGatherRows(torch.autograd.Function):
    """(B_local, E) -> (sum B_local, E), rank order.  backward: every rank holds the gradient of ITS OWN
    copy of the loss w.r.t. all rows; the gradient of the global objective (one copy of the loss, see
    global_train_loss) w.r.t. the local rows is the local slice -- no second collective needed."""

    @staticmethod
    def forward(ctx, x, sizes):
        world = dist.get_world_size()
        rank = dist.get_rank()
        outs = [x.new_empty((s,) + tuple(x.shape[1:])) for s in sizes]
        dist.all_gather(outs, x.contiguous())
        ctx.lo = sum(sizes[:rank])
        ctx.n = sizes[rank]
        return torch.cat(outs, dim=0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.lo:ctx.lo + ctx.n].contiguous(), None


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """Differentiable all-gather along dim 0 (ragged local sizes allowed)."""
    world = dist.get_world_size()
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    return _AllGatherRows.apply(x, [int(v.item()) for v in ns])


def all_gather_labels(labels: torch.Tensor) -> torch.Tensor:
    world = dist.get_world_size()
    n = torch.tensor([labels.shape[0]], dtype=torch.int64, device=labels.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    outs = [labels.new_empty((int(v.item()),)) for v in ns]
    dist.all_gather(outs, labels.contiguous())
    return torch.cat(outs)


def global_train_loss(loss_rec_local: torch.Tensor, n_local: int, n_global: int, loss_cl_global: torch.Tensor,
                      lambda_cl: float) -> torch.Tensor:
    """The scalar each rank back-propagates so that SUM-all-reduced gradients equal the gradient of the
    single-process loss  mean_global(rec) + lambda * InfoNCE(global batch)  (training.py:422):
      * the local MSE mean is re-weighted by its share of the global batch;
      * the InfoNCE term is computed identically on every rank from the gathered embeddings, and
        _AllGatherRows.backward hands each rank the slice that belongs to its own rows."""
    return loss_rec_local * (float(n_local) / float(n_global)) + lambda_cl * loss_cl_global


def allreduce_gradients(params: Iterable[torch.nn.Parameter]) -> None:
    """One flat fp32 SUM all-reduce over every gradient (a single bucket: the whole model is <= 12.6 MB, so
    the collective is latency-bound and bucketing would only add launches)."""
    ps: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    if not ps:
        return
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
