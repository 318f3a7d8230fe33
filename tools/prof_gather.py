#!/usr/bin/env python
"""The gather stage in isolation (north_star: "rocprof must show achieved HBM GB/s on the gather").

A 65 536-news token table (10 GB, MIND-small scale, S=50, D=768), one step's 28 160 news ids (B=512 x (50+5)),
uniform and Zipf(1.1):
  gather_rows : the standalone block copy (xnrs_gather_rows): algorithmic bytes = n x 153 600 B read + as much written
  qkv_gather  : the Q/K/V projection with the row gather folded into its A-operand loads (what forward_ids runs)
  qkv_dense   : the same projection on the materialised rows (reference point)
Prints wall-clock rates; run it under `rocprofv3 --kernel-trace --stats` and `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(separate passes) for the counter view:  python tools/prof_gather.py [uniform|zipf] [reps]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip, synth  # noqa: E402
from xnrs_amd.data import NewsStore  # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "zipf"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
n_news, S, D, n = 65536, 50, 768, 512 * 55
gen = torch.Generator(device=dev)
gen.manual_seed(31)
tx, tm = synth.device_tokens(gen, n_news + 1, S, D, dev)
store = NewsStore(tx, tm.reshape(n_news + 1, S), list(range(n_news)))
rng = np.random.default_rng(5)
if dist == "zipf":
    ids = np.minimum(rng.zipf(1.1, size=n), n_news)
else:
    ids = rng.integers(1, n_news + 1, size=n)
ids = torch.from_numpy(ids.astype(np.int32)).to(dev)
distinct = int(torch.unique(ids).numel())
w = torch.randn(3 * D, D, device=dev) / D ** 0.5


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


row_bytes = S * D * 4
t = timed(lambda: store.gather(ids))
print(f"[{dist}] {n} ids, {distinct} distinct: gather_rows {t*1e3:.3f} ms = {2 * n * (row_bytes + S * 4) / t / 1e9:.0f} GB/s "
      f"(read + write), {n * row_bytes / t / 1e9:.0f} GB/s of gathered rows", flush=True)
x, _ = store.gather(ids)
xt = tx.reshape(-1, D)
fl = 2.0 * n * S * 3 * D * D
y = torch.empty((n * S, 3 * D), dtype=torch.float32, device=dev)


def qkv(src, gids):
    hip.check(hip.lib().xnrs_linear_fwd(hip.ptr(src), hip.ptr(gids), S if gids is not None else 0, hip.ptr(w), None, hip.ptr(y),
                                        n * S, 3 * D, D, hip.ACT_NONE, hip.stream_ptr(dev)), "xnrs_linear_fwd")


tg = timed(lambda: qkv(xt, ids))
xd = x.reshape(-1, D)
td = timed(lambda: qkv(xd, None))
print(f"[{dist}] Q/K/V projection of the {n * S} gathered token rows: gather in the load {tg*1e3:.3f} ms ({fl/tg/1e12:.1f} TF, "
      f"{n * row_bytes / tg / 1e9:.0f} GB/s of gathered A rows)  dense rows {td*1e3:.3f} ms ({fl/td/1e12:.1f} TF)", flush=True)
