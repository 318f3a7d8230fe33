#!/usr/bin/env python
"""The attention-core kernels of the NRMS grad step on their own clock: news encoder forward + backward at the train shape
(1 600 history news x 50 x 768, 16 heads, half the slots empty, titles of 5..50 tokens, dropout 0.1), stage times from the
library's launch timer (stage 1 = attention forward, stage 9 = attention backward)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)
model, opt, batch, targets, labels, fn = bench.make_train_job("nrms", dev)
for rnd in range(3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    acc = {}
    for _ in range(5):
        hip.profile_enable((1 << 1) | (1 << 9))
        fn()
        torch.cuda.synchronize()
        st = hip.profile_read()
        hip.profile_enable(0)
        for k in ("attention_core", "bwd_attention_core"):
            acc.setdefault(k, []).append(st[k][0])
    print("  ".join(f"{k} {sorted(v)[len(v) // 2]:.3f} ms" for k, v in acc.items()), flush=True)
