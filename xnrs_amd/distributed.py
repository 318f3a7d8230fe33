"""Impression-parallel data parallelism over RCCL/xGMI (one process per GPU; backend "nccl" is RCCL
on ROCm, "gloo" on CPU for tests).

The reference has no distributed code at all; the path shards naturally by
impression/user: weights (0.66-3.15 M parameters) are replicated, every rank encodes and scores its own
impressions, and FORWARD/INFERENCE NEEDS NO COLLECTIVE.  The grad step (xnrs/training.py:402-431) adds
exactly two collectives and no host sync:
  1. ONE differentiable all-gather of [user embedding | theme-label bits] (gather_embeddings_and_labels), so the
     in-batch InfoNCE (training.py:433-472) sees the GLOBAL batch and equals the single-GPU loss; the shard sizes are
     exchanged once per run (ShardLayout), not per step;
  2. ONE flat fp32 all-reduce (SUM) of the gradients (2.6-12.6 MB: latency-bound -> a single bucket) in a persistent
     buffer the parameters' .grad are views of (GradBucket): no concatenation, no copy back.
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


class ShardLayout:
    """How many rows every rank contributes to the gathered batch.  Exchanged ONCE (one small all-gather and its host
    read, at set-up or whenever the local batch size changes) instead of before every payload collective; the step
    itself then knows every size on the host and never syncs.  Equal shards (the normal case: a fixed per-rank batch
    size, `drop_last`) gather straight into one tensor; a ragged last shard is padded to the largest shard for the
    collective and trimmed by host-known offsets."""

    def __init__(self, sizes):
        self.sizes = [int(v) for v in sizes]
        self.world = len(self.sizes)
        self.rank = dist.get_rank()
        self.max = max(self.sizes) if self.sizes else 0
        self.total = sum(self.sizes)
        self.equal = all(v == self.max for v in self.sizes)
        self.lo = sum(self.sizes[:self.rank])
        self.n_local = self.sizes[self.rank]

    @classmethod
    def exchange(cls, n_local: int, device=None):
        """The once-per-run size exchange (a collective + one host read: never inside the step)."""
        n = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
        ns = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
        dist.all_gather(ns, n)
        return cls(torch.cat(ns).tolist())

    @classmethod
    def uniform(cls, n_local: int):
        """Every rank holds n_local rows -- no communication at all (bench.py: fixed per-rank batch)."""
        return cls([int(n_local)] * dist.get_world_size())


def _gather_padded(x: torch.Tensor, layout: ShardLayout) -> torch.Tensor:
    """ONE collective: (n_local, W) -> (world * layout.max, W), rank-major, short shards zero-padded."""
    if x.shape[0] != layout.n_local:
        raise RuntimeError(f"local batch has {x.shape[0]} rows but the shard layout was exchanged for {layout.n_local}; "
                           "exchange a new ShardLayout when the per-rank batch size changes")
    if x.shape[0] < layout.max:
        x = torch.cat([x, x.new_zeros((layout.max - x.shape[0],) + tuple(x.shape[1:]))])
    out = x.new_empty((layout.world * layout.max,) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, x.contiguous())
    return out


def _trim(out: torch.Tensor, layout: ShardLayout) -> torch.Tensor:
    if layout.equal:
        return out
    return torch.cat([out[r * layout.max:r * layout.max + n] for r, n in enumerate(layout.sizes)])


class _GatherEmbeddingsAndLabels(torch.autograd.Function):
    """[user embedding | label bits] of every rank in ONE all-gather.  The int32 label travels as the bit pattern of an
    fp32 column next to the embedding (a collective copies bits; nothing computes on that column), so the labels cost
    no second collective.  backward: every rank holds the gradient of ITS OWN copy of the loss w.r.t. all rows; the
    gradient of the global objective (one copy of the loss, see global_train_loss) w.r.t. the local rows is the local
    slice -- no collective in the backward."""

    @staticmethod
    def forward(ctx, emb, labels, layout: ShardLayout):
        bits = labels.to(torch.int32).reshape(-1, 1).view(torch.float32)
        both = _trim(_gather_padded(torch.cat([emb.detach().to(torch.float32), bits], dim=1), layout), layout)
        ctx.layout = layout
        all_emb = both[:, :-1].contiguous()
        all_lab = both[:, -1:].contiguous().view(torch.int32).reshape(-1).to(labels.dtype)
        ctx.mark_non_differentiable(all_lab)
        return all_emb, all_lab

    @staticmethod
    def backward(ctx, g, _glab):
        lay = ctx.layout
        return g[lay.lo:lay.lo + lay.n_local].contiguous(), None, None


def gather_embeddings_and_labels(emb: torch.Tensor, labels: torch.Tensor, layout: ShardLayout):
    """(B_local,E) embeddings + (B_local,) integer labels -> ((sum B_local, E), (sum B_local,)) in rank order with ONE
    collective and no host sync; differentiable in `emb`.  Labels must fit int32 (theme / category ids do)."""
    return _GatherEmbeddingsAndLabels.apply(emb, labels, layout)


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layout: ShardLayout):
        ctx.layout = layout
        return _trim(_gather_padded(x.detach(), layout), layout)

    @staticmethod
    def backward(ctx, g):
        lay = ctx.layout
        return g[lay.lo:lay.lo + lay.n_local].contiguous(), None


def all_gather_rows(x: torch.Tensor, layout: ShardLayout = None) -> torch.Tensor:
    """Differentiable all-gather along dim 0.  With a ShardLayout: one collective, no host sync.  Without one the sizes
    are exchanged first (a second collective and a host read) -- set-up code and tests only."""
    return _AllGatherRows.apply(x, layout or ShardLayout.exchange(x.shape[0], x.device))


def all_gather_labels(labels: torch.Tensor, layout: ShardLayout = None) -> torch.Tensor:
    layout = layout or ShardLayout.exchange(labels.shape[0], labels.device)
    return _trim(_gather_padded(labels.reshape(-1, 1), layout), layout).reshape(-1)


def global_train_loss(loss_rec_local: torch.Tensor, n_local: int, n_global: int, loss_cl_global: torch.Tensor,
                      lambda_cl: float) -> torch.Tensor:
    """The scalar each rank back-propagates so that SUM-all-reduced gradients equal the gradient of the
    single-process loss  mean_global(rec) + lambda * InfoNCE(global batch)  (training.py:422):
      * the local MSE mean is re-weighted by its share of the global batch;
      * the InfoNCE term is computed identically on every rank from the gathered embeddings, and
        _AllGatherRows.backward hands each rank the slice that belongs to its own rows."""
    return loss_rec_local * (float(n_local) / float(n_global)) + lambda_cl * loss_cl_global


class GradBucket:
    """One persistent flat fp32 buffer that every parameter's `.grad` is a VIEW of: the backward accumulates straight
    into it, `allreduce()` is one in-place SUM all-reduce over it (a single bucket: the whole model is <= 12.6 MB, so the
    collective is latency-bound and bucketing would only add launches), and nothing is concatenated or copied back.
    Use `bucket.zero_grad()` instead of `optimizer.zero_grad()` (which would drop the views with set_to_none=True).

    One difference from the reference's `optimizer.zero_grad()` (set_to_none): EVERY parameter that requires grad holds a
    gradient view here, so a parameter the step never touches (dummy_param) carries zeros where the reference leaves None.
    Adam then keeps (zero) state for it and steps it by 0; an optimizer with weight decay WOULD decay it.  None of the
    shipped configs uses weight decay; pass only the parameters that take part in the step if yours does."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else None
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._attach()

    def _attach(self):
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def _attached(self) -> bool:
        off = 0
        for p in self.params:
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off or g.dtype != torch.float32:
                return False
            off += p.numel()
        return True

    def zero_grad(self):
        if not self._attached():  # someone replaced a .grad (optimizer.zero_grad(set_to_none=True), clip utilities ...)
            self._attach()
        self.flat.zero_()

    def allreduce(self):
        if not self._attached():
            # a gradient was re-allocated behind our back: fold what is there into the buffer, then re-attach
            off = 0
            for p in self.params:
                n = p.numel()
                g = p.grad
                if g is not None and g.data_ptr() != self.flat.data_ptr() + 4 * off:
                    self.flat[off:off + n].copy_(g.reshape(-1))
                off += n
            self._attach()
        if self.flat.numel():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


class OverlappedGradBuckets:
    """Several GradBuckets reduced AS THEY COMPLETE during the backward, on a side stream, overlapped with the rest of the
    backward (round 4; the single GradBucket reduces after the backward has finished).

    The gradients of a tower are complete when its last autograd node has run: in the grad step of the bi-encoders the
    user tower finishes first and the news tower (whose last node computes the Q|K|V weight gradients, the largest products
    of the step) last.  `groups` lists the parameters per bucket in that order -- `by_tower(model)` builds
    [user tower | scorer, news tower] for the ParentRec / NAML models.  Every parameter gets a post-accumulate hook; when
    the last parameter of a bucket has received its gradient the bucket's flat buffer is all-reduced asynchronously
    (communication stream; the compute stream waits for it only in `finish()`).  Buckets that never complete by hooks
    (a parameter without a gradient in this step, e.g. dummy_param) are reduced in `finish()`, so correctness never
    depends on the hooks: n_buckets collectives per step, always.

    Use: `buckets.zero_grad()`; forward; `loss.backward()`; `buckets.finish()`; `optimizer.step()`."""

    def __init__(self, groups):
        self.buckets = [GradBucket(g) for g in groups if any(p.requires_grad for p in g)]
        self._pending = []
        self._left = []
        self._work = []
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, _bi=bi: self._arrived(_bi)))
        self._reset()

    @classmethod
    def by_tower(cls, model: torch.nn.Module):
        """[everything that is not the news tower, the news tower]: the order in which their gradients complete."""
        news, rest = [], []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (news if name.startswith(("news_encoder.", "title_encoder.", "body_encoder.", "cat_", "subcat_", "feature_pooler."))
             else rest).append(p)
        return cls([g for g in (rest, news) if g])

    def _reset(self):
        self._left = [len(b.params) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._work = []

    def zero_grad(self):
        for b in self.buckets:
            b.zero_grad()
        self._reset()

    def _launch(self, bi):
        b = self.buckets[bi]
        self._launched[bi] = True
        if not b.flat.numel():
            return
        if not b._attached():
            b.allreduce()  # (a gradient was re-allocated behind the bucket's back: the synchronous, repairing path)
            return
        self._work.append(dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, async_op=True))

    def _arrived(self, bi):
        self._left[bi] -= 1
        if self._left[bi] == 0 and not self._launched[bi]:
            self._launch(bi)

    def finish(self):
        """After loss.backward(): reduce what the hooks did not, then make the current stream wait for every reduction."""
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        for w in self._work:
            w.wait()
        self._work = []

    allreduce = finish  # drop-in for GradBucket in a step that calls bucket.allreduce()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def allreduce_gradients(params: Iterable[torch.nn.Parameter]) -> None:
    """One-off flat SUM all-reduce (concatenates and copies back: tests and set-up code; the training loop keeps a
    GradBucket instead)."""
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
    """Every parameter and buffer from rank `src`.  The collective writes through `t.detach()` -- the same storage AND the same
    version counter as the parameter (a write through `t.data` would leave the counter where it was: the caches keyed on it,
    hip.folded_fc1 and the grad step's shared projections, would keep serving the old weights on the receiving ranks) -- and
    the caches are dropped as well."""
    from . import hip
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.detach(), src=src)
    hip.invalidate_fold_cache()
