#!/usr/bin/env python
"""A/B of one library knob on the NRMS grad step (bench.train_step_extra): `bench_train_ab.py XNRS_GEMM_DW 0 1`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import hip  # noqa: E402

knob, values = sys.argv[1], sys.argv[2:]
dev = torch.device("cuda", 0)
for rnd in range(2):
    for v in values:
        os.environ[knob] = v
        hip.reload_knobs()
        for name in ("NRMS", "standard"):
            r = bench.train_step_extra(dev, steps=10, warmup=3, model_name=name, variants=False)
            print(f"{knob}={v} {name}: {r['ms']:.3f} ms/step", flush=True)
