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

MAGIC = "xnrs_amd.newsstore.v2"
_MAGIC_V1 = "xnrs_amd.newsstore.v1"  # single text feature; still readable


class NewsStore:
    """x, m: the PRIMARY text feature's token table [n_rows,S,D] / mask [n_rows,S] (`feature`, 'title_emb' by default);
    `texts`: further text features over the same rows ({'abstract_emb': (x, m)}: NAML's second view, naml.py:76-80);
    `columns`: int32 categorical features per row ({'category_index': ...}, dataset.py:111-124)."""

    def __init__(self, x: torch.Tensor, m: torch.Tensor, ids: List, columns: Optional[Dict[str, torch.Tensor]] = None,
                 texts: Optional[Dict[str, tuple]] = None, feature: str = "title_emb"):
        assert x.dim() == 3 and m.shape == x.shape[:2]
        self.x, self.m = x, m  # row 0 = empty slot
        self.ids = list(ids)   # news id of row i+1
        self.index = {nid: i + 1 for i, nid in enumerate(self.ids)}
        self.columns = columns or {}
        self.feature = feature
        self.texts = dict(texts or {})
        for k, (tx, tm) in self.texts.items():
            assert tx.dim() == 3 and tm.shape == tx.shape[:2] and tx.shape[0] == x.shape[0], k
        self.pad_row = 0

    @property
    def n_rows(self) -> int:
        return self.x.shape[0]

    def text(self, feature: str):
        """(x, m) of a text feature."""
        if feature == self.feature:
            return self.x, self.m
        if feature not in self.texts:
            raise KeyError(f"news store has no text feature {feature!r} (has: {[self.feature] + sorted(self.texts)})")
        return self.texts[feature]

    def column(self, name: str) -> torch.Tensor:
        if name not in self.columns:
            raise KeyError(f"news store has no categorical column {name!r} (has: {sorted(self.columns)})")
        return self.columns[name]

    # ---- construction from the reference's in-memory dict (mind.py:161-164)
    @classmethod
    def from_news_feat(cls, news_feat: dict, feature: str = "title_emb", catg_features: Sequence[str] = (),
                       text_features: Sequence[str] = ()):
        """`text_features`: additional text features stored next to `feature` (e.g. ['abstract_emb'])."""
        ids = list(news_feat.keys())

        def table(feat):
            first = news_feat[ids[0]][feat]
            S, D = np.asarray(first[0]).shape[-2:]
            x = np.zeros((len(ids) + 1, S, D), dtype=np.float32)
            m = np.zeros((len(ids) + 1, S), dtype=np.float32)
            for i, nid in enumerate(ids):
                emb, mask = news_feat[nid][feat]
                x[i + 1] = np.asarray(emb, dtype=np.float32).reshape(S, D)
                m[i + 1] = np.asarray(mask, dtype=np.float32).reshape(S)
            return torch.from_numpy(x), torch.from_numpy(m)

        x, m = table(feature)
        texts = {f: table(f) for f in text_features if f != feature}
        cols = {}
        for f in catg_features:
            c = np.zeros((len(ids) + 1,), dtype=np.int32)  # pad label 0 (stack_scalars, xnrs/utils.py:66-73)
            for i, nid in enumerate(ids):
                c[i + 1] = int(news_feat[nid][f])
            cols[f] = torch.from_numpy(c)
        return cls(x, m, ids, cols, texts, feature)

    def to(self, device):
        return NewsStore(self.x.to(device), self.m.to(device), self.ids, {k: v.to(device) for k, v in self.columns.items()},
                         {k: (tx.to(device), tm.to(device)) for k, (tx, tm) in self.texts.items()}, self.feature)

    def rows(self, news_ids: Sequence) -> List[int]:
        return [self.index[n] for n in news_ids]

    def check_rows(self, rows: torch.Tensor, what: str = "rows") -> torch.Tensor:
        """int32, contiguous, on the table's device, every id inside [0, n_rows) -- a kernel never sees an id it would
        read out of bounds with (one host sync; callers on a hot loop validate their id arrays once and pass
        trusted=True to gather)."""
        if not rows.is_cuda or not self.x.is_cuda:
            raise hip.XnrsHipError(f"NewsStore: the table and the {what} must live on the HIP device")
        flat = rows.reshape(-1).to(torch.int32).contiguous()
        if flat.numel():
            lo, hi = int(flat.min().item()), int(flat.max().item())
            if lo < 0 or hi >= self.n_rows:
                raise IndexError(f"NewsStore: {what} out of range [{lo}, {hi}] for a table of {self.n_rows} rows")
        return flat

    def validate_masks(self) -> None:
        """Every mask of the store is 0 / 1 (one host sync per feature: set-up, not the step) -- what the padding-free
        encoders require and the device-compacted one cannot check without a sync (ops.check_binary_mask)."""
        from . import ops
        for name in [self.feature] + sorted(self.texts):
            ops.check_binary_mask(self.text(name)[1], f"NewsStore mask of '{name}'")

    def gather(self, rows: torch.Tensor, feature: Optional[str] = None, trusted: Optional[bool] = None):
        """Dense (x:(*rows.shape,S,D), m:(*rows.shape,S,1)) for int32 table rows -- the tensors the reference's dataset
        would have built on the host (dataset.py:63-85,97-109).  Only for consumers that need the batch itself (input
        gradients of the explainer); the encoders take the rows directly (forward_ids).

        Row ids are range-checked WITHOUT a host read by default (trusted=None): ids are clamped into the table, so the
        kernel never reads out of bounds, and an id that had to be clamped sets hip.STATUS_ROW_RANGE in the sticky device
        status word -- hip.check_status() raises at the caller's next sync point.  trusted=False: the old blocking check
        (two .item() reads, IndexError at once); trusted=True: no check at all."""
        tx, tm = self.text(feature or self.feature)
        tx, tm = hip.dev_f32(tx, "news table"), hip.dev_f32(tm, "news table mask")
        if trusted is False:
            flat = self.check_rows(rows)
        else:
            flat = rows.reshape(-1).to(torch.int32).contiguous()
            if trusted is None and flat.is_cuda and flat.numel():
                ok = flat.clamp(0, self.n_rows - 1)
                bad = (ok != flat).any().to(torch.int32) * hip.STATUS_ROW_RANGE
                hip.status_word(flat.device).bitwise_or_(bad.reshape(1))
                flat = ok
        if not flat.is_cuda:
            raise hip.XnrsHipError("NewsStore.gather: the rows must live on the HIP device")
        n, (S, D) = flat.numel(), tx.shape[1:]
        x = torch.empty((n, S, D), dtype=torch.float32, device=tx.device)
        m = torch.empty((n, S), dtype=torch.float32, device=tx.device)
        st = hip.stream_ptr(tx.device)
        hip.check(hip.lib().xnrs_gather_rows(hip.ptr(tx), hip.ptr(flat), hip.ptr(x), n, S * D, st), "xnrs_gather_rows(x)")
        hip.check(hip.lib().xnrs_gather_rows(hip.ptr(tm), hip.ptr(flat), hip.ptr(m), n, S, st), "xnrs_gather_rows(m)")
        return x.reshape(*rows.shape, S, D), m.reshape(*rows.shape, S, 1)

    def gather_column(self, name: str, rows: torch.Tensor) -> torch.Tensor:
        """Categorical feature of the given table rows (index bookkeeping: an int32 look-up)."""
        return self.column(name)[rows.long()]

    # ---- flat on-disk format (replaces the pandas pickle of mind.py:161-164):
    #   <path>.json                       header
    #   <path>.x.f32  <path>.m.u8         primary text feature  [n_rows,S,D] fp32 / [n_rows,S] u8
    #   <path>.<feat>.x.f32 / .m.u8       further text features
    #   <path>.<col>.i32                  categorical columns
    def _files(self, path: str):
        yield self.feature, path + ".x.f32", path + ".m.u8"
        for k in sorted(self.texts):
            yield k, f"{path}.{k}.x.f32", f"{path}.{k}.m.u8"

    def save(self, path: str, rows_per_chunk: int = 4096) -> None:
        header = {"magic": MAGIC, "n_rows": int(self.n_rows), "S": int(self.x.shape[1]), "D": int(self.x.shape[2]),
                  "ids": [str(i) for i in self.ids], "columns": sorted(self.columns), "feature": self.feature,
                  "texts": {k: {"S": int(tx.shape[1]), "D": int(tx.shape[2])} for k, (tx, tm) in sorted(self.texts.items())}}
        with open(path + ".json", "w") as f:
            json.dump(header, f)
        for feat, fx, fm in self._files(path):
            tx, tm = self.text(feat)
            with open(fx, "wb") as f:  # chunked: a 10-GB table never needs a second whole copy on the host
                for lo in range(0, self.n_rows, rows_per_chunk):
                    f.write(tx[lo:lo + rows_per_chunk].detach().cpu().numpy().astype(np.float32, copy=False).tobytes())
            tm.detach().cpu().numpy().astype(np.uint8).tofile(fm)
        for k, v in self.columns.items():
            v.detach().cpu().numpy().astype(np.int32).tofile(f"{path}.{k}.i32")

    @staticmethod
    def _open(path: str):
        """Header + read-only memory maps of every payload file (nothing is read yet)."""
        with open(path + ".json") as f:
            h = json.load(f)
        if h.get("magic") not in (MAGIC, _MAGIC_V1):
            raise ValueError(f"{path}.json is not a {MAGIC} header")
        n = h["n_rows"]
        feats = [(h.get("feature", "title_emb"), path + ".x.f32", path + ".m.u8", h["S"], h["D"])]
        for k, sd in sorted(h.get("texts", {}).items()):
            feats.append((k, f"{path}.{k}.x.f32", f"{path}.{k}.m.u8", sd["S"], sd["D"]))
        maps = {}
        for feat, fx, fm, S, D in feats:
            if os.path.getsize(fx) != n * S * D * 4 or os.path.getsize(fm) != n * S:
                raise ValueError(f"news store payload size does not match its header ({feat})")
            maps[feat] = (np.memmap(fx, dtype=np.float32, mode="r", shape=(n, S, D)),
                          np.memmap(fm, dtype=np.uint8, mode="r", shape=(n, S)))
        cols = {}
        for k in h["columns"]:
            if os.path.getsize(f"{path}.{k}.i32") != n * 4:
                raise ValueError(f"news store column {k} does not match its header")
            cols[k] = torch.from_numpy(np.fromfile(f"{path}.{k}.i32", dtype=np.int32))
        return h, feats, maps, cols

    @classmethod
    def load(cls, path: str, mmap: bool = True):
        """Host copy of a saved store.  mmap=True: the tensors are VIEWS of read-only memory maps (pages are read on
        first touch, no whole-table buffer is allocated); mmap=False reads the files into RAM.  To put a table into HBM
        use load_to_device -- it never holds more than one chunk on the host."""
        h, feats, maps, cols = cls._open(path)

        def host(feat):
            mx, mm = maps[feat]
            if mmap:
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")  # "non-writable array": the store is read-only by contract
                    return torch.from_numpy(mx), torch.from_numpy(np.asarray(mm)).float()
            return torch.from_numpy(np.array(mx)), torch.from_numpy(np.array(mm)).float()

        primary = feats[0][0]
        x, m = host(primary)
        texts = {f[0]: host(f[0]) for f in feats[1:]}
        return cls(x, m, h["ids"], cols, texts, primary)

    @classmethod
    def load_to_device(cls, path: str, device, rows_per_chunk: int = 2048, stats: Optional[dict] = None):
        """File -> HBM without a whole-table host copy (replaces the pandas pickle load of
        xnrs/data/mind.py:161-164): the device table is allocated once, the file is memory-mapped, and chunks of
        `rows_per_chunk` rows travel through TWO reused pinned staging buffers (the copy of chunk i overlaps the page-in
        of chunk i+1) straight into their rows of the table.  The u8 mask is widened to the fp32 0/1 mask the kernels
        read after it has landed (one elementwise cast of S bytes per row: data movement, not hot-path arithmetic).
        `stats` (optional dict) receives bytes / seconds / GB/s."""
        import time
        dev = torch.device(device)
        if dev.type != "cuda":
            raise hip.XnrsHipError("NewsStore.load_to_device: the target must be a HIP device")
        h, feats, maps, cols = cls._open(path)
        n = h["n_rows"]
        t0 = time.perf_counter()
        nbytes = 0
        copy_stream = torch.cuda.Stream(device=dev)
        out = {}
        for feat, fx, fm, S, D in feats:
            mx, mm = maps[feat]
            tx = torch.empty((n, S, D), dtype=torch.float32, device=dev)
            rpc = max(1, min(int(rows_per_chunk), n))
            pinned = [torch.empty((rpc, S, D), dtype=torch.float32).pin_memory() for _ in range(2)]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            used = [False, False]
            with torch.cuda.stream(copy_stream):
                for i, lo in enumerate(range(0, n, rpc)):
                    hi = min(lo + rpc, n)
                    b = i & 1
                    if used[b]:
                        done[b].synchronize()  # the previous copy out of this staging buffer has landed
                    np.copyto(pinned[b][:hi - lo].numpy(), mx[lo:hi])  # page-in: file -> pinned buffer
                    tx[lo:hi].copy_(pinned[b][:hi - lo], non_blocking=True)
                    done[b].record(copy_stream)
                    used[b] = True
                mu8 = torch.from_numpy(np.array(mm)).pin_memory()
                tm = mu8.to(dev, non_blocking=True).to(torch.float32)
            nbytes += mx.nbytes + mm.nbytes
            out[feat] = (tx, tm)
        copy_stream.synchronize()
        torch.cuda.current_stream(dev).wait_stream(copy_stream)
        dt = time.perf_counter() - t0
        if stats is not None:
            stats.update(bytes=int(nbytes), seconds=dt, gb_per_s=nbytes / dt / 1e9, rows=int(n), rows_per_chunk=int(rows_per_chunk))
        primary = feats[0][0]
        x, m = out.pop(primary)
        return cls(x, m, h["ids"], {k: v.to(dev) for k, v in cols.items()}, out, primary)


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
