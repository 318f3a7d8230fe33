#!/usr/bin/env python
"""Accuracy (against an fp64 product) and speed of the three forward-GEMM modes, interleaved in one process.

    python tools/bench_split.py [M N K ...]     # on the GPU box
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip, ops  # noqa: E402

dev = torch.device("cuda", 0)
args = [int(v) for v in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(65500, 2304, 768), (65500, 768, 768), (65500, 256, 768),
                                                                    (30720, 900, 300), (4099, 260, 300)]
names = {0: "f32", 1: "bf16x3", 2: "bf16x2"}
torch.manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    rows = slice(0, min(M, 4096))
    ref = (x[rows].double() @ w.double().t() + b.double())
    scale = ref.abs().max().item()
    err, res = {}, {m: [] for m in names}
    for rnd in range(5):
        for m in names:
            hip.set_gemm_mode(m)
            y = ops.linear(x, w, b)
            if rnd == 0:
                err[m] = ((y[rows].double() - ref).abs().max().item() / scale, torch.isfinite(y).all().item())
                if M > 4096:  # the tail rows too (edge tiles)
                    t = slice(M - 300, M)
                    rt = x[t].double() @ w.double().t() + b.double()
                    err[m] = (max(err[m][0], (y[t].double() - rt).abs().max().item() / scale), err[m][1])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.linear(x, w, b)
            e1.record()
            torch.cuda.synchronize()
            res[m].append(e0.elapsed_time(e1) / 10)
    hip.set_gemm_mode(0)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: " + "  ".join(
        f"{names[m]}: {fl/sorted(t)[len(t)//2]/1e9:.1f} TF err {err[m][0]:.2e}{'' if err[m][1] else ' NONFINITE'}" for m, t in res.items()), flush=True)
