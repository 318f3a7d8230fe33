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


@torch.no_grad()
def encode_news_table(model, store: NewsStore, rows_per_call: int = 0):
    """All rows of the store through the model's news tower -> (vecs:(n_rows,E), hm:(n_rows,1)).  Row 0 (the
    empty slot) gets the tower's output for an all-padded news (head-bias leak, SURVEY.md finding 4).

    Works for every model on the path through its `encode_news_ids(store, ids)` hook (ParentRec: one TextEncoder over
    the title table; NAML: title + abstract tables and the two category columns, naml.py:76-107).  `rows_per_call`
    bounds the rows handed to one hook call (0 = all at once; the C ABI chunks its own workspace either way)."""
    n = store.n_rows
    step = n if rows_per_call <= 0 else int(rows_per_call)
    ys, hms = [], []
    for lo in range(0, n, step):
        ids = torch.arange(lo, min(lo + step, n), dtype=torch.int32, device=store.x.device).reshape(1, -1)
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


@torch.no_grad()
def evaluate(model, store: NewsStore, behaviors: Behaviors, l_hist: int, batch: int = 4096) -> Dict[str, float]:
    """Mean of the per-impression metrics over all sessions (training.py:245-303 aggregates the same way)."""
    vecs, hm = encode_news_table(model, store)
    batcher = DeviceBatcher(behaviors, l_hist, store.pad_row)
    dev = vecs.device
    sums = torch.zeros(len(METRIC_NAMES), dtype=torch.float64, device=dev)
    n = len(behaviors)
    for lo in range(0, n, batch):
        sess = torch.arange(lo, min(lo + batch, n), device=dev)
        hist, off, rows, csess, targets = batcher.eval_batch(sess)
        h = vecs[hist.long()]            # (B, l_hist, E) row gather of pre-encoded vectors (data movement only)
        m = hm[hist.long()]
        u = model.encode_user(h, m)
        r = score_csr(vecs, rows, csess, u, relu=True)
        sums += rank_metrics(r, targets, off).double().sum(0)
    return {k: float(v) / n for k, v in zip(METRIC_NAMES, sums.tolist())}
