"""CPU oracle for the data-side rows.  TEST INFRASTRUCTURE ONLY (same rules as
oracle/xnrs_oracle.py): numpy / plain-Python restatements of

* NewsRecDataset.__getitem__ + custom_collate_fn  (xnrs/data/dataset.py:48-163, xnrs/utils.py:190-204)
* the per-impression metrics                        (xnrs/evaluation/metrics.py:9-64, training.py:210-227)

pinned by tests/golden/data.npz, which tests/golden/make_golden.py records from the REAL reference classes.
"""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def mix64(seed: int, a: int, b: int) -> int:
    """The counter-based draw of xnrs_assemble_train_batch (splitmix64 finaliser), bit for bit."""
    z = (seed + 0x9E3779B97F4A7C15 * ((a * 0x100000001B3 + b + 1) & MASK64)) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def history_rows(history_rows_all, l_hist, pad_row=0):
    """dataset.py:77-85: the LAST l_hist clicks, zero padding behind."""
    h = list(history_rows_all)[-l_hist:]
    return h + [pad_row] * (l_hist - len(h))


def train_rows(session_index, hist, pos, neg, l_hist, n_neg, seed, pad_row=0):
    """dataset.py:54-57 with the counter-based draws: one positive (random.choice), n_neg negatives with
    replacement (random.choices)."""
    p = pos[mix64(seed, session_index, 0) % len(pos)] if len(pos) else pad_row
    n = [neg[mix64(seed, session_index, c) % len(neg)] if len(neg) else pad_row for c in range(1, n_neg + 1)]
    return history_rows(hist, l_hist, pad_row), [p] + n


def eval_rows(hist, pos, neg, l_hist, pad_row=0):
    """dataset.py:58-61,149: all positives then all negatives; targets 1.. then 0.."""
    return history_rows(hist, l_hist, pad_row), list(pos) + list(neg), [1.0] * len(pos) + [0.0] * len(neg)


def materialise(x_table, m_table, rows):
    """What the reference hands the model for these rows: (N,S,D) tokens, (N,S,1) mask (dataset.py:78-85,97-109)."""
    rows = np.asarray(rows, dtype=np.int64)
    return x_table[rows], m_table[rows][..., None]


# ------------------------------------------------------------------------------ metrics (metrics.py:9-64)
def _order(y_score):
    """np.argsort(y_score)[::-1] with ties broken 'higher original index first' (stable sort reversed)."""
    return np.argsort(np.asarray(y_score), kind="stable")[::-1]


def impression_metrics(y_true, y_score):
    """-> [ndcg@5, ndcg@10, rr, ctr@1, ctr@10, auc, acc, rec, prec] for one impression, after the
    np.nan_to_num of training.py:210-211."""
    t = np.asarray(y_true, dtype=np.float64)
    s = np.nan_to_num(np.asarray(y_score, dtype=np.float64), nan=0.0, posinf=1.0, neginf=0.0)
    order = _order(s)

    def dcg(yt, od, k):
        y = np.take(yt, od[:k])
        return np.sum((2 ** y - 1) / np.log2(np.arange(len(y)) + 2))

    def ndcg(k):
        return dcg(t, order, k) / dcg(t, _order(t), k)

    ranked = np.take(t, order)
    rr = np.max(ranked / (np.arange(len(ranked)) + 1))
    ctr1, ctr10 = np.mean(ranked[:1]), np.mean(ranked[:10])
    pos, neg = s[t > 0.5], s[t <= 0.5]
    auc = (np.sum(pos[:, None] > neg[None, :]) + 0.5 * np.sum(pos[:, None] == neg[None, :])) / (len(pos) * len(neg))
    pred = np.round(np.clip(s, 0, 1))
    tp, fp = np.sum((pred == 1) & (t == 1)), np.sum((pred == 1) & (t == 0))
    fn, tn = np.sum((pred == 0) & (t == 1)), np.sum((pred == 0) & (t == 0))
    acc = (tp + tn) / len(t)
    rec = tp / (tp + fn)
    prec = tp / (tp + fp) if (tp + fp) > 0 else 0.0
    return np.array([ndcg(5), ndcg(10), rr, ctr1, ctr10, auc, acc, rec, prec])
