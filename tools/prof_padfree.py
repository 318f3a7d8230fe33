#!/usr/bin/env python
"""Stage times of one benchmark step in the padding-free modes (hipEvent stage profile of the C ABI)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
model, _ = bench.build_model(w, dev)
hist, cand = bench.make_inputs(w, dev, seed=1000)
enc = model.news_encoder
with torch.no_grad():
    for name, flags in (("dense", {}), ("skip_empty", dict(skip_empty=True)), ("unpadded", dict(unpadded=True)),
                        ("both", dict(skip_empty=True, unpadded=True))):
        for k, v in flags.items():
            setattr(enc, k, v)
        for _ in range(3):
            bench.step(model, hist, cand)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            bench.step(model, hist, cand)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5 * 1e3
        hip.profile_enable(hip.PROFILE_ALL)
        bench.step(model, hist, cand)
        torch.cuda.synchronize()
        st = hip.profile_read()
        hip.profile_enable(0)
        enc.skip_empty = enc.unpadded = False
        print(f"{name:11s} wall {wall:7.3f} ms | " + " ".join(f"{k}={v[0]:.3f}" for k, v in st.items()) +
              f" | sum {sum(v[0] for v in st.values()):.3f}", flush=True)
