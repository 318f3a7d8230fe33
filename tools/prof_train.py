#!/usr/bin/env python
"""N train steps of NRMS (B=64, H=25, C=5, S=50, D=768) for rocprofv3 --kernel-trace."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
r = bench.train_step_extra(dev, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3, warmup=1,
                           model_name=sys.argv[2] if len(sys.argv) > 2 else "nrms", variants=False)
print(r)
