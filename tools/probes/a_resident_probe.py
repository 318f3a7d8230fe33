#!/usr/bin/env python
"""Probe: is the N = 256 forward GEMM held back by WHERE its A operand comes from?  The same 65 536 x 256 x 768 product
with the A rows gathered (a) from a table as large as the product (row i -> row i: streams 201 MB) and (b) from a
4 096-row table (row i -> i % 4096: 12.6 MB, L2-resident after the first touch).  Same kernel (gathered-A variant), same
FLOPs; N = 2304 beside it."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xnrs_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)
K, M = 768, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator(device=dev)
g.manual_seed(1)
x = torch.randn(M, K, device=dev, generator=g)
l = hip.lib()


def clock(fn, reps=20, warm_s=0.5):
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
    y = torch.empty(M, N, device=dev)
    for name, ids in (("identity ids (A streams)", torch.arange(M, dtype=torch.int32, device=dev)),
                      ("ids % 4096 (A resident)", (torch.arange(M, device=dev) % 4096).to(torch.int32)),
                      ("ids % 512  (A resident)", (torch.arange(M, device=dev) % 512).to(torch.int32))):
        def fn():
            hip.check(l.xnrs_linear_fwd(hip.ptr(x), hip.ptr(ids), 1, hip.ptr(w), None, hip.ptr(y), M, N, K, hip.ACT_NONE,
                                        hip.stream_ptr(dev)), "xnrs_linear_fwd")
        dt = clock(fn)
        print(f"N {N:5d} {name:28s} {dt * 1e3:7.3f} ms  {2.0 * M * N * K / dt / 1e12:6.1f} TF  {2.0 * M * N * K / dt / 1e12 / 157.3:.3f}", flush=True)
