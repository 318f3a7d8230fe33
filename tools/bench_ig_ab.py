#!/usr/bin/env python
"""IG loop (explain.py:160-166, one impression per step) with and without the skipped weight gradients
(autograd.SKIP_UNUSED_DW), interleaved in one process; and the batched form of xnrs_amd.explain."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import autograd as AG  # noqa: E402

dev = torch.device("cuda", 0)
for name in ("standard", "NRMS"):
    for H in (9, 25):
        w = dict(B=1, H=H, C=1, S=50, D=768, h=16, E=256, A=256)
        model, _ = bench.build_model(w, dev, model_name=name)
        (hx, hm), (cx, cm) = bench.make_inputs(w, dev, seed=50 + H, full_history=True)
        hx = hx.clone().requires_grad_()

        def run(n):
            c, _ = model.news_encoder((cx, cm))
            for a in torch.arange(1.0 / n, 1 + 1.0 / n, 1.0 / n)[:n]:
                ga = a * hx
                ha, ham = model.news_encoder((ga, hm))
                ua = model.user_encoder.forward(inpt=(ha, ham))
                sa = torch.relu(model.rec_model(ua, c))
                g = torch.autograd.grad(sa, ga)[0]
            return g
        res = {True: [], False: []}
        for rnd in range(3):
            for skip in (True, False):
                AG.SKIP_UNUSED_DW = skip
                run(5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(60)
                torch.cuda.synchronize()
                res[skip].append(60 / (time.perf_counter() - t0))
        AG.SKIP_UNUSED_DW = True
        print(f"{name} H={H}: it/s with the unused dW skipped {max(res[True]):.0f} ({', '.join(f'{v:.0f}' for v in res[True])}); "
              f"computed {max(res[False]):.0f} ({', '.join(f'{v:.0f}' for v in res[False])})", flush=True)
