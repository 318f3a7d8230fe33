#!/usr/bin/env python
"""Probe: does the forward GEMM's efficiency depend on the operands' ROW STRIDE (K floats: 3 072 B at K = 768 -- rows of a
tile's 64-byte k chunk land 12 x 256 B apart, a handful of L2 channels)?  Same M, N; K = 768 and neighbours."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xnrs_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M = 65536
g = torch.Generator(device=dev)
g.manual_seed(1)


def clock(fn, reps=20, warm_s=0.4):
    t_end = time.perf_counter() + warm_s
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for N in (256, 2304):
    for K in (768, 772, 784, 800, 832, 1024, 1040):
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        dt = clock(lambda: ops.linear(x, w, None))
        print(f"N {N:5d} K {K:5d} (row stride {K * 4:5d} B): {dt * 1e3:7.3f} ms  {2.0 * M * N * K / dt / 1e12:6.1f} TF  "
              f"{2.0 * M * N * K / dt / 1e12 / 157.3:.3f}", flush=True)
        del x, w
