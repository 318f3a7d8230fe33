#!/usr/bin/env python
"""What does the attention dropout (counter RNG per probability, forward and backward) cost in the NRMS grad step?
The same step with p = 0.1 and p = 0 (outputs not shared, so both run both encodes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import autograd as AG, hip  # noqa: E402
from xnrs_amd.models.components import layers  # noqa: E402

dev = torch.device("cuda", 0)
AG.SHARE_OUTPUTS = False
model, opt, batch, targets, labels, fn = bench.make_train_job("nrms", dev)
for p in (0.1, 0.0, 0.1, 0.0):
    for mod in model.modules():
        if isinstance(mod, layers.MultiHeadAttention):
            mod.dropout.p = p
    dt = bench.timed(fn, 20, 5, False) / 20
    hip.profile_enable(hip.PROFILE_ALL)
    fn()
    torch.cuda.synchronize()
    st = hip.profile_read()
    hip.profile_enable(0)
    print(f"p={p}: {dt * 1e3:.3f} ms/step; attention fwd {st['attention_core'][0]:.3f} ms, bwd {st['bwd_attention_core'][0]:.3f} ms", flush=True)
