#!/usr/bin/env python
"""Grad step (NRMS / StandardRec / NAML, reference call order, B = 64) with the backward's side lane on and off, interleaved
in one process (XNRS_BWD_SIDE_STREAM through xnrs_reload_knobs).  Also checks that both settings give the same parameters
after a step from the same state (bitwise: the same launches, only their streams differ)."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)


def set_lane(on, min_rows=0):
    os.environ["XNRS_BWD_SIDE_STREAM"] = "1" if on else "0"
    os.environ["XNRS_BWD_SIDE_MIN_ROWS"] = str(min_rows)
    hip.lib().xnrs_reload_knobs()


for name in sys.argv[1:] or ["nrms", "standard", "naml"]:
    model, opt, batch, targets, labels, fn = bench.make_train_job(name, dev)
    for _ in range(3):
        fn()
    res = {}
    for rnd in range(3):
        for on in (True, False):
            set_lane(on)
            dt = bench.timed(fn, 20, 3, False) / 20
            res.setdefault(on, []).append(dt * 1e3)
    # same gradients either way
    gr = {}
    for on in (True, False):
        set_lane(on)
        opt.zero_grad()
        torch.manual_seed(5)
        fn(step_opt=False)
        torch.cuda.synchronize()
        gr[on] = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    same = all(torch.equal(a, b) for a, b in zip(gr[True], gr[False]))
    md = max(float((a - b).abs().max()) for a, b in zip(gr[True], gr[False]))
    if name == "nrms":  # which towers carry the gain: every attention tower / not the user tower (1 600 rows) / only the history (80 000)
        for mr in (0, 4096, 40000):
            set_lane(True, mr)
            print(f"  nrms, lane for towers of >= {mr} rows: {bench.timed(fn, 20, 3, False) / 20 * 1e3:.3f} ms", flush=True)
    set_lane(True)
    print(f"{name:9s} side lane on  {min(res[True]):7.3f} ms (runs {', '.join(f'{v:.3f}' for v in res[True])})   "
          f"off {min(res[False]):7.3f} ms (runs {', '.join(f'{v:.3f}' for v in res[False])})   grads equal {same} (max diff {md:.2e})", flush=True)
