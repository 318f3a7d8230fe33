#!/usr/bin/env python
"""Time the one-launch additive encoder (additive_fused.hip) alone: n_news x S x D through xnrs_text_encoder_fwd with the
kernel forced, one library per process (XNRS_LIB=libxnrs_hip_afexpN.so = a diagnostic build with parts switched off).

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
for mode, label in (("2", "fused"), ("0", "pipeline")):
    with hip.knobs(XNRS_ADDITIVE_FUSED=mode), torch.no_grad():
        fn = lambda: ops.text_encoder_forward(x, m, None, enc.pooler, None)  # noqa: E731
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print(f"{os.environ.get('XNRS_LIB', 'libxnrs_hip.so'):28s} {label:9s} {n} x {S} x {D} (A={A}): {dt * 1e3:7.3f} ms  {fl / dt / 1e12:6.1f} TF  {fl / dt / 1e12 / 157.3:.3f}")
    if os.environ.get("XNRS_LIB"):
        break

if not os.environ.get("XNRS_LIB"):  # the practical ceiling in this harness: the plain fc1 GEMM (no epilogue work) of the same shape
    w = enc.pooler.fc1.weight
    xf = x.reshape(n * S, D)
    with torch.no_grad():
        for _ in range(5):
            ops.linear(xf, w, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.linear(xf, w, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    fg = 2.0 * n * S * D * A
    print(f"{'plain GEMM ' + str(n * S) + ' x ' + str(A) + ' x ' + str(D):58s} {dt * 1e3:7.3f} ms  {fg / dt / 1e12:6.1f} TF  {fg / dt / 1e12 / 157.3:.3f}")
