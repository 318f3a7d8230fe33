#!/usr/bin/env python
"""PMC driver for the N = 256 question: ops.linear at (M, N) = (655 360, 256), (65 536, 256) and (65 500, 2304), K = 768, five
launches each, so `rocprofv3 --pmc FETCH_SIZE` / TCC counters show what each A panel costs in fabric reads against the
algorithmic M x K x 4 bytes (tools/gpu_pmc_any.sh <tag> probes/n256_traffic.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xnrs_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
K = 768
g = torch.Generator(device=dev)
g.manual_seed(1)
xbig = torch.randn(655360, K, device=dev, generator=g)
for (M, N) in [(655360, 256), (65536, 256), (65500, 2304)]:
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    for _ in range(5):
        y = ops.linear(xbig[:M], w, None)
    torch.cuda.synchronize()
    print(M, N, "alg read MB", (M * K + N * K) * 4 / 1e6, "write MB", M * N * 4 / 1e6, flush=True)
