#!/usr/bin/env python
"""A/B of the additive-only models' forward (StandardRec, NAML at B=512, H=25, C=5, S=50, D=768) under library knobs,
interleaved in one process:  python tools/bench_other_models.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests.test_hip_naml_ids import big_batch, big_model  # noqa: E402
from xnrs_amd import hip  # noqa: E402

VARIANTS = [("pipeline, ocml tanh", dict(XNRS_ADDITIVE_FUSED="0", XNRS_FAST_TANH="0")),
            ("pipeline, fast tanh", dict(XNRS_ADDITIVE_FUSED="0", XNRS_FAST_TANH="1")),
            ("fused, fast tanh, 1 fragment set", dict(XNRS_ADDITIVE_FUSED="1", XNRS_FAST_TANH="1", XNRS_AF_FBUF="1")),
            ("fused, fast tanh, 2 fragment sets", dict(XNRS_ADDITIVE_FUSED="1", XNRS_FAST_TANH="1", XNRS_AF_FBUF="2"))]
for name in ("standard", "NAML"):
    model, _ = big_model(name)
    batch = big_batch(name, 512, 25, 5)
    fl = bench.other_model_flops(name, 25, 5) * 512
    with torch.no_grad():
        for rep in range(2):
            for label, kn in VARIANTS:
                with hip.knobs(**kn):
                    fn = lambda: model(batch)  # noqa: E731
                    dt = bench.timed(fn, 8, 3, False) / 8
                if rep == 1:
                    print(f"{name:9s} {label:36s} {dt * 1e3:7.3f} ms  {512 / dt:9.0f} impr/s  {fl / dt / 1e12:6.1f} TF  {fl / dt / 1e12 / 157.3:.3f}")
