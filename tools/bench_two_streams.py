#!/usr/bin/env python
"""Does the news encoder gain from running two halves of a batch on two HIP streams (the HBM-bound attention core / pooling
of one half behind the MFMA-bound projections of the other)?  25 600 news x 50 x 768 through NRMS's news encoder."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
model, _ = bench.build_model(w, dev)
hist, cand = bench.make_inputs(w, dev, seed=1000)
enc = model.news_encoder
x = hist[0].reshape(-1, w["S"], w["D"])
m = hist[1].reshape(-1, w["S"], 1)
n = x.shape[0]


def clock(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run_split(k, prio=False):
    # prio: the streams alternate between high and normal priority (a high-priority queue's workgroups take freed slots first)
    streams = [torch.cuda.Stream(priority=(-1 if (prio and i % 2 == 0) else 0)) for i in range(k)]
    bounds = [n * i // k for i in range(k + 1)]
    outs = [None] * k

    def fn():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = ops.text_encoder(x[bounds[i]:bounds[i + 1]], m[bounds[i]:bounds[i + 1]], enc)
        for s in streams:
            cur.wait_stream(s)
    return fn, outs


with torch.no_grad():
    ref = ops.text_encoder(x, m, enc)
    t1 = clock(lambda: ops.text_encoder(x, m, enc))
    print(f"one stream: {t1:.3f} ms", flush=True)
    for prio in (False, True):
        for k in (2, 3, 4):
            fn, outs = run_split(k, prio)
            tk = clock(fn)
            y = torch.cat([o[0] for o in outs])
            print(f"{k} streams{' (alternating priority)' if prio else ''}: {tk:.3f} ms  ({t1 / tk:.3f}x)  equal {torch.equal(y, ref[0])}", flush=True)
