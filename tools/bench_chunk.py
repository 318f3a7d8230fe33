#!/usr/bin/env python
"""News-encoder time over one benchmark step's 28 160 news for different pass sizes (sequences per chunk)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
model, _ = bench.build_model(w, dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
n = 28160
x, m = synth.device_tokens(gen, n, w["S"], w["D"], dev)
x, m = x.reshape(n, w["S"], w["D"]), m.reshape(n, w["S"], 1)
chunks = [int(v) for v in sys.argv[1:]] or [1310, 655, 2620, 5240, 1310]
res = {i: [] for i in range(len(chunks))}
with torch.no_grad():
    for rnd in range(4):
        for i, c in enumerate(chunks):
            for _ in range(2):
                ops.text_encoder(x, m, model.news_encoder, chunk=c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ops.text_encoder(x, m, model.news_encoder, chunk=c)
            torch.cuda.synchronize()
            res[i].append((time.perf_counter() - t0) / 5 * 1e3)
for i, c in enumerate(chunks):
    print(f"chunk {c:6d} news: {sorted(res[i])[len(res[i]) // 2]:.3f} ms")
