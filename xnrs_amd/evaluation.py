"""Evaluation path on the device: the reference's test loop runs batch_size 1 and
re-encodes every candidate of every impression (xnrs/training.py:61-67,194-243); here every news is
encoded ONCE per epoch, impressions are scored as CSR candidate lists against those vectors, and the
per-impression metrics of xnrs/evaluation/metrics.py are computed by a HIP kernel."""
from __future__ import annotations

from typing import Dict

import torch

from . import hip
from .data import Behaviors, DeviceBatcher, NewsStore

METRIC_NAMES = ("ndcg@5", "ndcg@10", "rr", "ctr@1", "ctr@10", "auc", "acc", "rec", "prec")


def _dist_rank_world(distributed):
    """(rank, world, on) of the evaluation job: `distributed` None = the default process group when one with more than one
    rank exists, False = this process alone, True = the default group (must be initialised)."""
    import torch.distributed as dist
    if distributed is False or not (dist.is_available() and dist.is_initialized()):
        if distributed is True:
            raise RuntimeError("evaluate(distributed=True) needs an initialised torch.distributed process group")
        return 0, 1, False
    world = dist.get_world_size()
    return dist.get_rank(), world, world > 1


def sharded_mean(local_sums: torch.Tensor, n_total: int, on: bool) -> torch.Tensor:
    """Sum of the per-rank metric sums (ONE all-reduce, fp64) over the global number of impressions."""
    if on:
        import torch.distributed as dist
        dist.all_reduce(local_sums, op=dist.ReduceOp.SUM)
    return local_sums / max(int(n_total), 1)


@torch.no_grad()
def encode_news_table(model, store: NewsStore, rows_per_call: int = 0, rows=None):
    """All rows of the store through the model's news tower -> (vecs:(n_rows,E), hm:(n_rows,1)).  Row 0 (the
    empty slot) gets the tower's output for an all-padded news (head-bias leak, SURVEY.md finding 4).

    Works for every model on the path through its `encode_news_ids(store, ids)` hook (ParentRec: one TextEncoder over
    the title table; NAML: title + abstract tables and the two category columns, naml.py:76-107).  `rows_per_call`
    bounds the rows handed to one hook call (0 = all at once; the C ABI chunks its own workspace either way)."""
    r0, r1 = (0, store.n_rows) if rows is None else rows  # (rows: the slice of the table this rank encodes)
    n = r1 - r0
    step = max(n, 1) if rows_per_call <= 0 else int(rows_per_call)
    ys, hms = [], []
    for lo in range(r0, max(r1, r0 + 1), step):
        ids = torch.arange(lo, max(min(lo + step, r1), lo), dtype=torch.int32, device=store.x.device).reshape(1, -1)
        y, hm = model.encode_news_ids(store, ids)
        ys.append(y[0])
        hms.append(hm[0])
    return (ys[0], hms[0]) if len(ys) == 1 else (torch.cat(ys), torch.cat(hms))


def score_csr(vecs: torch.Tensor, cand_rows: torch.Tensor, cand_sess: torch.Tensor, u: torch.Tensor, relu: bool = True):
    vecs = hip.dev_f32(vecs, "news vectors")
    u = hip.dev_f32(u, "user vectors").reshape(-1, vecs.shape[1])
    r = torch.empty((cand_rows.numel(),), dtype=torch.float32, device=vecs.device)
    hip.check(hip.lib().xnrs_score_csr(hip.ptr(vecs), hip.ptr(cand_rows), hip.ptr(cand_sess), hip.ptr(u), hip.ptr(r),
                                       cand_rows.numel(), vecs.shape[1], int(relu), hip.stream_ptr(vecs.device)), "xnrs_score_csr")
    return r


def rank_metrics(scores: torch.Tensor, targets: torch.Tensor, cand_off: torch.Tensor):
    """-> (B, 9) tensor in METRIC_NAMES order (xnrs/evaluation/metrics.py:9-64 per impression)."""
    B = cand_off.numel() - 1
    out = torch.empty((B, len(METRIC_NAMES)), dtype=torch.float32, device=scores.device)
    hip.check(hip.lib().xnrs_rank_metrics(hip.ptr(hip.dev_f32(scores, "scores")), hip.ptr(hip.dev_f32(targets, "targets")),
                                          hip.ptr(cand_off), hip.ptr(out), B, hip.stream_ptr(scores.device)), "xnrs_rank_metrics")
    return out


def encode_news_table_sharded(model, store: NewsStore, rank: int, world: int):
    """Every rank encodes a contiguous slice of the table (the 6.6 TFLOP of a MIND-large corpus divide by the world size) and
    ONE all-gather hands every rank all vectors: rows padded to the largest slice, [vector | news mask] side by side."""
    import torch.distributed as dist
    from .distributed import shard_range
    lo, hi = shard_range(store.n_rows, rank, world)
    y, hm = encode_news_table(model, store, rows=(lo, hi)) if hi > lo else (None, None)
    sizes = [shard_range(store.n_rows, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    dev = store.x.device
    E = model.encode_news_ids(store, torch.zeros((1, 1), dtype=torch.int32, device=dev))[0].shape[-1] if y is None else y.shape[-1]
    mine = torch.zeros((mx, E + 1), dtype=torch.float32, device=dev)
    if y is not None:
        mine[:hi - lo, :E] = y
        mine[:hi - lo, E:] = hm.reshape(-1, 1)
    out = torch.empty((world * mx, E + 1), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, mine)
    both = torch.cat([out[r * mx:r * mx + (b - a)] for r, (a, b) in enumerate(sizes)])
    return both[:, :E].contiguous(), both[:, E:].contiguous()


@torch.no_grad()
def evaluate(model, store: NewsStore, behaviors: Behaviors, l_hist: int, batch: int = 4096, distributed=None) -> Dict[str, float]:
    """Mean of the per-impression metrics over all sessions (training.py:245-303 aggregates the same way).

    With a torch.distributed process group of more than one rank (`distributed` None / True) the epoch is rank-sharded: the
    news table is encoded in slices (one all-gather of the vectors), every rank scores a contiguous block of the sessions
    (distributed.shard_range) and ONE fp64 all-reduce adds the metric sums -- every rank returns the same dict, equal to the
    single-process result up to the order of that final sum.  (The reference's test loop is one process, batch size 1:
    training.py:194-243.)"""
    rank, world, on = _dist_rank_world(distributed)
    if on:
        from .distributed import shard_range
        vecs, hm = encode_news_table_sharded(model, store, rank, world)
        s_lo, s_hi = shard_range(len(behaviors), rank, world)
    else:
        vecs, hm = encode_news_table(model, store)
        s_lo, s_hi = 0, len(behaviors)
    batcher = DeviceBatcher(behaviors, l_hist, store.pad_row)
    dev = vecs.device
    sums = torch.zeros(len(METRIC_NAMES), dtype=torch.float64, device=dev)
    n = len(behaviors)
    for lo in range(s_lo, s_hi, batch):
        sess = torch.arange(lo, min(lo + batch, s_hi), device=dev)
        hist, off, rows, csess, targets = batcher.eval_batch(sess)
        h = vecs[hist.long()]            # (B, l_hist, E) row gather of pre-encoded vectors (data movement only)
        m = hm[hist.long()]
        u = model.encode_user(h, m)
        r = score_csr(vecs, rows, csess, u, relu=True)
        sums += rank_metrics(r, targets, off).double().sum(0)
    mean = sharded_mean(sums, n, on)
    out = {k: float(v) for k, v in zip(METRIC_NAMES, mean.tolist())}
    hip.check_status(dev)  # (the read above was the epoch's sync point: what the sync-free encoders could not raise, raises here)
    return out
