#!/usr/bin/env python
"""Small profiling driver: N passes of the NRMS news encoder over one chunk of news (default 1310
news x 50 tokens x 768 = one 65 500-row pass), so rocprofv3 sees each kernel N times.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/prof_news.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import synth  # noqa: E402

n_news = int(sys.argv[1]) if len(sys.argv) > 1 else 1310
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S, D, h = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (50, 768, 16)
dev = torch.device("cuda", 0)
w = dict(B=1, H=1, C=1, S=S, D=D, h=h, E=256 if D % 15 else 240, A=256)
model, _ = bench.build_model(w, dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
x, m = synth.device_tokens(gen, n_news, S, D, dev)
x, m = x.reshape(1, n_news, S, D), m.reshape(1, n_news, S, 1)
with torch.no_grad():
    for _ in range(passes):
        y, hm = model.news_encoder((x, m))
    torch.cuda.synchronize()
print("ok", tuple(y.shape), float(y.abs().mean()))
