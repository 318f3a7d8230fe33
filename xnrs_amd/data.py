"""Device-resident data path: what replaces NewsRecDataset.__getitem__ +
custom_collate_fn + the `.to(device)` inside the encoders (xnrs/data/dataset.py:48-163, xnrs/utils.py:190-204)
once the encoders are fast -- the host gather/pad/cat and the 4.6 MB/impression PCIe copy.

* NewsStore      : flat [n_rows, S, D] fp32 token table + [n_rows, S] mask (+ optional category columns), row 0
                   is the EMPTY SLOT (all-zero tokens and mask: the zero padding of dataset.py:82-85); built from
                   the reference's in-memory format (mind.py:161-164: {news_id: {feat: (emb(1,S,D), mask(1,S))}}),
                   saved to / memory-mapped from a flat file (replaces the pandas pickle).
* Behaviors      : click histories / positives / negatives as CSR arrays of table rows, on the device.
* DeviceBatcher  : train / eval batches as ROW IDS, assembled by HIP kernels; the model consumes them through
                   ParentRec.forward_ids (id gather fused into the first GEMM's load).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import hip

MAGIC = "xnrs_amd.newsstore.v1"


class NewsStore:
    def __init__(self, x: torch.Tensor, m: torch.Tensor, ids: List, columns: Optional[Dict[str, torch.Tensor]] = None):
        assert x.dim() == 3 and m.shape == x.shape[:2]
        self.x, self.m = x, m  # row 0 = empty slot
        self.ids = list(ids)   # news id of row i+1
        self.index = {nid: i + 1 for i, nid in enumerate(self.ids)}
        self.columns = columns or {}
        self.pad_row = 0

    # ---- construction from the reference's in-memory dict (mind.py:161-164)
    @classmethod
    def from_news_feat(cls, news_feat: dict, feature: str = "title_emb", catg_features: Sequence[str] = ()):
        ids = list(news_feat.keys())
        first = news_feat[ids[0]][feature]
        S, D = np.asarray(first[0]).shape[-2:]
        x = np.zeros((len(ids) + 1, S, D), dtype=np.float32)
        m = np.zeros((len(ids) + 1, S), dtype=np.float32)
        for i, nid in enumerate(ids):
            emb, mask = news_feat[nid][feature]
            x[i + 1] = np.asarray(emb, dtype=np.float32).reshape(S, D)
            m[i + 1] = np.asarray(mask, dtype=np.float32).reshape(S)
        cols = {}
        for f in catg_features:
            c = np.zeros((len(ids) + 1,), dtype=np.int32)  # pad label 0 (stack_scalars, xnrs/utils.py:66-73)
            for i, nid in enumerate(ids):
                c[i + 1] = int(news_feat[nid][f])
            cols[f] = torch.from_numpy(c)
        return cls(torch.from_numpy(x), torch.from_numpy(m), ids, cols)

    def to(self, device):
        return NewsStore(self.x.to(device), self.m.to(device), self.ids, {k: v.to(device) for k, v in self.columns.items()})

    def rows(self, news_ids: Sequence) -> List[int]:
        return [self.index[n] for n in news_ids]

    def gather(self, rows: torch.Tensor):
        """Dense (x:(*rows.shape,S,D), m:(*rows.shape,S,1)) for int32 table rows -- the tensors the reference's dataset
        would have built on the host (dataset.py:63-85,97-109).  Only for consumers that need the batch itself (input
        gradients of the explainer); the encoders take the rows directly (forward_ids)."""
        if not self.x.is_cuda or not rows.is_cuda:
            raise hip.XnrsHipError("NewsStore.gather: the table and the rows must live on the HIP device")
        flat = rows.reshape(-1).to(torch.int32).contiguous()
        n, (S, D) = flat.numel(), self.x.shape[1:]
        x = torch.empty((n, S, D), dtype=torch.float32, device=self.x.device)
        m = torch.empty((n, S), dtype=torch.float32, device=self.x.device)
        st = hip.stream_ptr(self.x.device)
        hip.check(hip.lib().xnrs_gather_rows(hip.ptr(self.x), hip.ptr(flat), hip.ptr(x), n, S * D, st), "xnrs_gather_rows(x)")
        hip.check(hip.lib().xnrs_gather_rows(hip.ptr(self.m), hip.ptr(flat), hip.ptr(m), n, S, st), "xnrs_gather_rows(m)")
        return x.reshape(*rows.shape, S, D), m.reshape(*rows.shape, S, 1)

    # ---- flat on-disk format: <path>.json (header) + <path>.x.f32 + <path>.m.u8 (+ <path>.<col>.i32)
    def save(self, path: str) -> None:
        x = self.x.detach().cpu().numpy()
        m = self.m.detach().cpu().numpy()
        header = {"magic": MAGIC, "n_rows": int(x.shape[0]), "S": int(x.shape[1]), "D": int(x.shape[2]),
                  "ids": [str(i) for i in self.ids], "columns": sorted(self.columns)}
        with open(path + ".json", "w") as f:
            json.dump(header, f)
        x.astype(np.float32).tofile(path + ".x.f32")
        m.astype(np.uint8).tofile(path + ".m.u8")
        for k, v in self.columns.items():
            v.detach().cpu().numpy().astype(np.int32).tofile(f"{path}.{k}.i32")

    @classmethod
    def load(cls, path: str, mmap: bool = True):
        with open(path + ".json") as f:
            h = json.load(f)
        if h.get("magic") != MAGIC:
            raise ValueError(f"{path}.json is not a {MAGIC} header")
        n, S, D = h["n_rows"], h["S"], h["D"]
        if os.path.getsize(path + ".x.f32") != n * S * D * 4 or os.path.getsize(path + ".m.u8") != n * S:
            raise ValueError("news store payload size does not match its header")
        if mmap:
            x = np.memmap(path + ".x.f32", dtype=np.float32, mode="r", shape=(n, S, D))
            m = np.memmap(path + ".m.u8", dtype=np.uint8, mode="r", shape=(n, S))
        else:
            x = np.fromfile(path + ".x.f32", dtype=np.float32).reshape(n, S, D)
            m = np.fromfile(path + ".m.u8", dtype=np.uint8).reshape(n, S)
        cols = {k: torch.from_numpy(np.fromfile(f"{path}.{k}.i32", dtype=np.int32)) for k in h["columns"]}
        # np.array(...) copies out of the read-only mapping page by page (no second whole-file buffer on top of
        # the page cache); the caller then moves the tensors to the device once
        return cls(torch.from_numpy(np.array(x)), torch.from_numpy(np.array(m)).float(), h["ids"], cols)


def _csr(lists: List[List[int]]):
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum([len(l) for l in lists], out=off[1:])
    val = np.fromiter((v for l in lists for v in l), dtype=np.int32, count=int(off[-1]))
    return torch.from_numpy(off), torch.from_numpy(val)


class Behaviors:
    """Sessions (xnrs/data/dataset.py:50-51: {'history', 'positives', 'negatives', 'main_theme', ...}) as CSR
    arrays of table rows."""

    def __init__(self, hist, pos, neg, themes: List[str]):
        (self.hist_off, self.hist_val), (self.pos_off, self.pos_val), (self.neg_off, self.neg_val) = hist, pos, neg
        self.themes = themes
        uniq = sorted(set(themes))
        self.theme_labels = torch.tensor([uniq.index(t) for t in themes], dtype=torch.int64)

    @classmethod
    def from_sessions(cls, sessions: Sequence[dict], store: NewsStore):
        hist = _csr([store.rows(s["history"]) for s in sessions])
        pos = _csr([store.rows(s["positives"]) for s in sessions])
        neg = _csr([store.rows(s["negatives"]) for s in sessions])
        return cls(hist, pos, neg, [str(s.get("main_theme", "")) for s in sessions])

    def __len__(self):
        return self.hist_off.numel() - 1

    def to(self, device):
        b = Behaviors.__new__(Behaviors)
        for k in ("hist_off", "hist_val", "pos_off", "pos_val", "neg_off", "neg_val", "theme_labels"):
            setattr(b, k, getattr(self, k).to(device))
        b.themes = self.themes
        return b


class DeviceBatcher:
    def __init__(self, behaviors: Behaviors, l_hist: int, pad_row: int = 0):
        if not behaviors.hist_off.is_cuda:
            raise hip.XnrsHipError("Behaviors must live on the HIP device (behaviors.to('cuda'))")
        self.b, self.l_hist, self.pad_row = behaviors, int(l_hist), int(pad_row)

    def _common(self, sess):
        if not sess.is_cuda:
            raise hip.XnrsHipError("session indices must live on the HIP device")
        b = self.b
        return (hip.ptr(sess), sess.numel(), hip.ptr(b.hist_off), hip.ptr(b.hist_val), hip.ptr(b.pos_off), hip.ptr(b.pos_val),
                hip.ptr(b.neg_off), hip.ptr(b.neg_val))

    def train_batch(self, sess: torch.Tensor, n_neg: int, seed: int):
        """-> hist_rows:(B,l_hist) int32, cand_rows:(B,1+n_neg) int32, targets:(B,1+n_neg,1) [1,0,..]"""
        sess = sess.to(torch.int64).contiguous()
        B, dev = sess.numel(), sess.device
        hist = torch.empty((B, self.l_hist), dtype=torch.int32, device=dev)
        cand = torch.empty((B, 1 + n_neg), dtype=torch.int32, device=dev)
        hip.check(hip.lib().xnrs_assemble_train_batch(*self._common(sess), self.l_hist, n_neg, self.pad_row, int(seed),
                                                      hip.ptr(hist), hip.ptr(cand), hip.stream_ptr(dev)),
                  "xnrs_assemble_train_batch")
        targets = torch.zeros((B, 1 + n_neg, 1), dtype=torch.float32, device=dev)
        targets[:, 0] = 1.0
        return hist, cand, targets

    def eval_batch(self, sess: torch.Tensor):
        """-> hist_rows:(B,l_hist), cand_off:(B+1) int64, cand_rows:(n,), cand_sess:(n,), targets:(n,)"""
        sess = sess.to(torch.int64).contiguous()
        B, dev = sess.numel(), sess.device
        b = self.b
        counts = (b.pos_off[sess + 1] - b.pos_off[sess]) + (b.neg_off[sess + 1] - b.neg_off[sess])
        off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=off[1:])
        n = int(off[-1].item())
        hist = torch.empty((B, self.l_hist), dtype=torch.int32, device=dev)
        rows = torch.empty((n,), dtype=torch.int32, device=dev)
        csess = torch.empty((n,), dtype=torch.int32, device=dev)
        targets = torch.empty((n,), dtype=torch.float32, device=dev)
        hip.check(hip.lib().xnrs_assemble_eval_batch(*self._common(sess), self.l_hist, self.pad_row, hip.ptr(off), hip.ptr(hist),
                                                     hip.ptr(rows), hip.ptr(csess), hip.ptr(targets), hip.stream_ptr(dev)),
                  "xnrs_assemble_eval_batch")
        return hist, off, rows, csess, targets
