#!/usr/bin/env python
"""Benchmark step with the padding-free encoder, device-compacted (default) or host-compacted (argv[1] == "host"):
wall time per step, for a kernel trace around it (tools/gpu_trace.sh)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "device"
ops.COMPACT_ON_DEVICE = mode != "host"
dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
model, _ = bench.build_model(w, dev)
hist, cand = bench.make_inputs(w, dev, seed=1000)
enc = model.news_encoder
enc.unpadded = True
enc.skip_empty = len(sys.argv) > 2 and sys.argv[2] == "skip"
with torch.no_grad():
    for _ in range(3):
        bench.step(model, hist, cand)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        bench.step(model, hist, cand)
    torch.cuda.synchronize()
    print(f"{mode} skip_empty={enc.skip_empty}: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per step", flush=True)
