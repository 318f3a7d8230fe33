#!/usr/bin/env python
"""Time the one-launch additive encoder (additive_fused.hip) alone: n_news x S x D through xnrs_text_encoder_fwd with the
kernel forced, against the GEMM + pooling pipeline and, for scale, the plain fc1 GEMM and a Q/K/V-shaped GEMM of the same
kernel family -- every figure after 0.6 s of back-to-back calls (settled clocks).  One library per process
(XNRS_LIB=libxnrs_hip_afexpN.so = a diagnostic build with parts switched off, `make -C xnrs_amd/csrc afexp`).

    python tools/bench_af.py [n_news S D A]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip  # noqa: E402

if os.environ.get("XNRS_LIB"):
    hip.LIB_PATH = os.path.join(ROOT, "xnrs_amd", os.environ["XNRS_LIB"])
from xnrs_amd import ops, synth  # noqa: E402
from xnrs_amd.models.components import layers, news_encoding  # noqa: E402

n, S, D, A = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (12800, 50, 768, 256)
dev = torch.device("cuda", 0)
enc = news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=D, in_features=D, head=False, att=None)
enc.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, 3))
enc = enc.eval().to(dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
x, m = synth.device_tokens(gen, n, S, D, dev)
fl = n * (2.0 * S * D * A + 2.0 * S * (A + D))
LIB = os.environ.get("XNRS_LIB", "libxnrs_hip.so")


def clock(fn, reps=20, warm_s=0.6):
    """Mean seconds per call after `warm_s` seconds of back-to-back calls (the chip's clocks settle under the load)."""
    t_end = time.perf_counter() + warm_s
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def line(label, dt, flops):
    print(f"{LIB:26s} {label:44s} {dt * 1e3:7.3f} ms  {flops / dt / 1e12:6.1f} TF  {flops / dt / 1e12 / 157.3:.3f}", flush=True)


with torch.no_grad():
    for mode, label in (("2", "fused"), ("0", "pipeline")):
        with hip.knobs(XNRS_ADDITIVE_FUSED=mode):
            dt = clock(lambda: ops.text_encoder_forward(x, m, None, enc.pooler, None))
        line(f"{label}: {n} news x {S} x {D}, A={A}", dt, fl)
        if os.environ.get("XNRS_LIB"):
            break
    if not os.environ.get("XNRS_LIB"):
        w = enc.pooler.fc1.weight
        xf = x.reshape(n * S, D)
        dt = clock(lambda: ops.linear(xf, w, None))
        line(f"plain GEMM {n * S} x {A} x {D}", dt, 2.0 * n * S * D * A)
        rows = 65500  # the Q/K/V projection of one pass: the same kernel, 18 column tiles per row tile
        wq = torch.randn(3 * D, D, device=dev) / D ** 0.5
        xq = xf[:rows].contiguous()
        dt = clock(lambda: ops.linear(xq, wq, None))
        line(f"plain GEMM {rows} x {3 * D} x {D}", dt, 2.0 * rows * D * 3 * D)
