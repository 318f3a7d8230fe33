#!/usr/bin/env python
"""Profiling driver for BASELINE configs[3]/[4]: N forward passes of StandardRec or NAML at B=512, H=25, C=5, S=50, D=768
so rocprofv3 sees every kernel N times.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/prof_other_models.py standard|NAML [passes]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_hip_naml_ids import big_batch, big_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "standard"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model, _ = big_model(name)
batch = big_batch(name, 512, 25, 5)
with torch.no_grad():
    for _ in range(passes):
        r = model(batch)
    torch.cuda.synchronize()
print("ok", name, tuple(r.shape), float(r.abs().mean()))
