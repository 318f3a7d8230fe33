#!/usr/bin/env python
"""NRMS grad step under the Python-side switches of xnrs_amd.autograd (stage times from the library's launch timer)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import autograd as AG, hip  # noqa: E402

dev = torch.device("cuda", 0)
model, opt, batch, targets, labels, fn = bench.make_train_job("nrms", dev)
for name, kw in [("default", {}), ("host lists", dict(DEVICE_LISTS=False)), ("dense K|V", dict(KV_ROWS=False)),
                 ("no lists", dict(LIVE_ROWS=False)), ("default", {})]:
    old = {k: getattr(AG, k) for k in kw}
    for k, v in kw.items():
        setattr(AG, k, v)
    try:
        dt = bench.timed(fn, 15, 4, False) / 15
        hip.profile_enable(hip.PROFILE_ALL)
        fn()
        torch.cuda.synchronize()
        st = hip.profile_read()
        hip.profile_enable(0)
    finally:
        for k, v in old.items():
            setattr(AG, k, v)
    print(f"{name:12s} {dt * 1e3:7.3f} ms/step  " + "  ".join(f"{k} {v[0]:.3f}ms/{v[2] / max(v[0], 1e-9) / 1e9:.0f}TF" for k, v in st.items() if v[1]), flush=True)
