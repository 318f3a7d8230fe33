#!/usr/bin/env python
"""Robustness: a 4x benchmark batch (B = 2048: 112 640 news, 5.6 M token rows, 17 GB of tokens) must give, impression by
impression, bitwise the scores of the same impressions run in B = 512 pieces (64-bit offsets, chunk loop, grid limits)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
w["B"] = 2048
model, _ = bench.build_model(w, dev)
hist, cand = bench.make_inputs(w, dev, seed=77)
with torch.no_grad():
    big = bench.step(model, hist, cand)
    torch.cuda.synchronize()
    ok = True
    for b0 in range(0, w["B"], 512):
        sl = slice(b0, b0 + 512)
        part = bench.step(model, (hist[0][sl], hist[1][sl]), (cand[0][sl], cand[1][sl]))
        ok = ok and torch.equal(part, big[sl])
    t = bench.timed(lambda: bench.step(model, hist, cand), 3, 1, False) / 3
print(f"B=2048: scores finite {bool(torch.isfinite(big).all())}, equal to the B=512 pieces {ok}, {w['B'] / t:.0f} impressions/s")
sys.exit(0 if ok else 1)
