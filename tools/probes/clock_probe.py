#!/usr/bin/env python
"""Which clock does the chip hold under the N = 256 product and under the N = 2304 projection?  Runs each GEMM back to back
for a few seconds and prints wall-clock windows; tools/gpu_clock_probe.sh samples rocm-smi beside it and joins the two."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xnrs_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
K = 768
g = torch.Generator(device=dev)
g.manual_seed(1)
xbig = torch.randn(655360, K, device=dev, generator=g)
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
for (M, N) in [(65500, 2304), (655360, 256), (65500, 2304), (655360, 256)]:
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    x = xbig[:M]
    t0 = time.time()
    n = 0
    while time.time() - t0 < SEC:
        for _ in range(20):
            ops.linear(x, w, None)
        torch.cuda.synchronize()
        n += 20
    t1 = time.time()
    # the last second only (settled)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(40):
        ops.linear(x, w, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - ta) / 40
    print(f"WINDOW {t0:.3f} {time.time():.3f} M {M} N {N} {2.0 * M * N * K / dt / 1e12:.1f} TF {2.0 * M * N * K / dt / 1e12 / 157.3:.3f}", flush=True)
    time.sleep(1.0)
