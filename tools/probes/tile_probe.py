#!/usr/bin/env python
"""Probe: the N = 256 forward GEMM (65 536 x 256 x 768) under every block tile (XNRS_GEMM_TILE = 0..3)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xnrs_amd import hip, ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
M, K = 65536, 768
x = torch.randn(M, K, device=dev, generator=g)


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
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    for tile in ("-1", "0", "1", "2", "3"):
        with hip.knobs(XNRS_GEMM_TILE=tile):
            dt = clock(lambda: ops.linear(x, w, None))
        print(f"N {N:5d} tile {tile:>2s}: {dt * 1e3:7.3f} ms  {2.0 * M * N * K / dt / 1e12 / 157.3:.3f}", flush=True)
