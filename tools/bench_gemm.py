#!/usr/bin/env python
"""A/B micro-benchmark of the fp32 MFMA GEMM variants (interleaved rounds in ONE process).

    python tools/bench_gemm.py            # on the GPU box
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip, ops  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [(65500, 2304, 768), (65500, 768, 768), (30720, 960, 320)]
if os.environ.get("XNRS_BENCH_SHAPES"):  # "M,N,K;M,N,K"
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["XNRS_BENCH_SHAPES"].split(";")]
# variants: "name:ENV=v+ENV=v,name2:..."; default compares the plain double-buffered pipeline with the default one
spec = sys.argv[1] if len(sys.argv) > 1 else "p1k32:XNRS_GEMM_PIPE=1+XNRS_GEMM_BK=32,default:,p5k32:XNRS_GEMM_PIPE=5+XNRS_GEMM_BK=32,p5k16_3wg:XNRS_GEMM_PIPE=5+XNRS_GEMM_BK=16"
variants = {}
for item in spec.split(","):
    name, _, envs = item.partition(":")
    variants[name] = dict(e.split("=") for e in envs.split("+") if e)
ALL_KEYS = ("XNRS_GEMM_PIPE", "XNRS_GEMM_BK", "XNRS_GEMM_BUF", "XNRS_GEMM_GROUP", "XNRS_GEMM_TILE")
torch.manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ref = None
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v, env in variants.items():
            for k in ALL_KEYS:
                os.environ.pop(k, None)
            os.environ.update({k: e for k, e in env.items() if k != "MODE"})
            hip.reload_knobs()  # the library reads its knobs at load and on request only
            hip.set_gemm_mode(int(env.get("MODE", 0)))  # "MODE=1|2": the bf16-split kernels
            y = ops.linear(x, w, b)  # warm
            if rnd == 0:
                if ref is None:
                    ref = y.clone()
                elif "MODE" not in env:
                    assert torch.equal(ref, y), f"variant {v} differs"
                else:
                    assert (ref - y).abs().max() <= 1e-4 * ref.abs().max(), f"variant {v} differs"
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear(x, w, b)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: " + "  ".join(
        f"{v}: {fl/sorted(t)[len(t)//2]/1e9:.1f} TF" for v, t in res.items()))
