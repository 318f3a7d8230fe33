#!/usr/bin/env python
"""MultiHeadAttention forward+backward time with the fused / two-kernel attention backward (XNRS_MHA_BWD_FUSED)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip  # noqa: E402
from xnrs_amd.models.components import layers  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [(1600, 50, 768, 16), (4096, 30, 320, 16), (4096, 30, 300, 15), (3200, 25, 256, 16), (4096, 16, 256, 8)]
for (n, S, D, h) in shapes:
    torch.manual_seed(0)
    att = layers.MultiHeadAttention(h, D).to(dev).eval()
    x = torch.randn(n, S, D, device=dev, requires_grad=True)
    m = (torch.rand(n, S, 1, device=dev) < 0.8).float()
    w = torch.randn(n, S, D, device=dev)
    res = {}
    for rnd in range(3):
        for flag in ("1", "0"):
            os.environ["XNRS_MHA_BWD_FUSED"] = flag
            hip.reload_knobs()  # the library reads its knobs at load and on request only
            for _ in range(2):
                att.zero_grad(set_to_none=True)
                (att(x, m) * w).sum().backward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                att.zero_grad(set_to_none=True)
                (att(x, m) * w).sum().backward()
            torch.cuda.synchronize()
            res.setdefault(flag, []).append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"n={n} S={S} D={D} h={h}: fused {sorted(res['1'])[1]:.3f} ms  two-kernel {sorted(res['0'])[1]:.3f} ms", flush=True)
