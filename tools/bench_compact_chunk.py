#!/usr/bin/env python
"""Device-compacted encoder: ms per call against the pass size (news per pass) at two shares of empty news."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
w = dict(bench.WORKLOAD)
model, _ = bench.build_model(w, dev)
enc = model.news_encoder
n, S, D = 25600, w["S"], w["D"]
gen = torch.Generator(device=dev)
gen.manual_seed(5)
x, m = synth.device_tokens(gen, n, S, D, dev)


def clock(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.no_grad():
    for empty in (0.0, 0.5, 0.95):
        keep = (torch.rand(n, 1, 1, device=dev, generator=gen) >= empty).float()
        xe, me = x * keep, m * keep
        ref = ops.text_encoder(xe, me, enc)[0]
        for chunk in (1310, 2620, 5240, 10480, 25600):
            y = ops.text_encoder_forward_compact(xe, me, enc.att, enc.pooler, enc.head, chunk=chunk)[0]
            t = clock(lambda: ops.text_encoder_forward_compact(xe, me, enc.att, enc.pooler, enc.head, chunk=chunk))
            print(f"empty {empty:.2f} news/pass {chunk:6d}: {t:7.3f} ms  equal {torch.equal(y, ref)}", flush=True)
