#!/usr/bin/env python
"""The grad-step extras of bench.py on their own (NRMS / StandardRec / NAML at B = 64, and the IG step)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["nrms", "standard", "naml", "ig"]
for name in which:
    if name == "ig":
        r = bench.ig_step_extra(dev)
    else:
        r = bench.train_step_extra(dev, steps=10, warmup=3, model_name=name)
    print(name, json.dumps(r), flush=True)
