#!/usr/bin/env python
"""Where does the N = 256 forward GEMM lose against N = 2304?  ops.linear (plain fp32 GEMM, no activation) over M rows x K = 768
for several (M, N): workgroup rounds, grid sizes below / at / above one full round (1024 workgroups of 128 x 128), settled clocks."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
K = 768
g = torch.Generator(device=dev)
g.manual_seed(1)
xbig = torch.randn(700000, K, device=dev, generator=g)


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


for (M, N) in [(65500, 2304), (65536, 256), (57344, 256), (49152, 256), (32768, 256), (131072, 256), (262144, 256), (655360, 256),
               (65536, 512), (65536, 768), (65536, 1536), (32768, 2304), (16384, 2304)]:
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    x = xbig[:M]
    dt = clock(lambda: ops.linear(x, w, None))
    wgs = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M {M:7d} N {N:5d}: {dt * 1e3:7.3f} ms  {2.0 * M * N * K / dt / 1e12:6.1f} TF  {2.0 * M * N * K / dt / 1e12 / 157.3:.3f}   "
          f"{wgs} workgroups = {wgs / 1024:.2f} rounds", flush=True)
