#!/usr/bin/env python
"""Device-compacted encoder over 25 600 news of which a given share is empty (argv[1], default 0.95): a few calls for a
kernel trace -- what do the worst-case grids cost when almost every workgroup leaves at once?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import ops, synth  # noqa: E402

empty = float(sys.argv[1]) if len(sys.argv) > 1 else 0.95
dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
model, _ = bench.build_model(w, dev)
enc = model.news_encoder
gen = torch.Generator(device=dev)
gen.manual_seed(5)
x, m = synth.device_tokens(gen, 25600, w["S"], w["D"], dev)
keep = (torch.rand(25600, 1, 1, device=dev, generator=gen) >= empty).float()
x, m = x * keep, m * keep
with torch.no_grad():
    for _ in range(4):
        y = ops.text_encoder_unpadded(x, m, enc)
    torch.cuda.synchronize()
print("ok", float(y[0].abs().mean()))
