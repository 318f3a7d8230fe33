#!/usr/bin/env python
"""Device-compacted encoder (xnrs_text_encoder_fwd_compact) against the share of empty news: ms per call at the benchmark's
news shape, beside the dense call and the host-compacted one (one sync)."""
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
enc = model.news_encoder
n, S, D = 25600, w["S"], w["D"]
gen = torch.Generator(device=dev)
gen.manual_seed(5)
x, m = synth.device_tokens(gen, n, S, D, dev)


def clock(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.no_grad():
    for empty in (0.0, 0.25, 0.5, 0.75, 0.95):
        keep = (torch.rand(n, 1, 1, device=dev, generator=gen) >= empty).float()
        xe, me = x * keep, m * keep
        live_rows = int(me.ne(0).sum())
        t_dense = clock(lambda: ops.text_encoder(xe, me, enc))
        ops.COMPACT_ON_DEVICE = True
        t_dev = clock(lambda: ops.text_encoder_unpadded(xe, me, enc))
        ops.COMPACT_ON_DEVICE = False
        t_host = clock(lambda: ops.text_encoder_unpadded(xe, me, enc))
        print(f"empty {empty:.2f} live rows {live_rows / (n * S):.3f}: dense {t_dense:7.3f} ms  device-compacted {t_dev:7.3f} ms  "
              f"host-compacted (no news skip) {t_host:7.3f} ms", flush=True)
