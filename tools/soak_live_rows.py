#!/usr/bin/env python
"""Soak: the row lists of the grad step (live token rows, token rows of the non-empty news) against the dense step over
random shapes -- every attention kernel family (S <= 32 head-per-wave, 33..64 pair, > 64 generic; fused and two-kernel
backward), with and without an attention tower, with and without the id table, empty news always present.
Forward outputs must be bitwise equal, gradients within 2e-5 (summation order).   python tools/soak_live_rows.py [n] [seed0]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import autograd as AG  # noqa: E402
from xnrs_amd import synth  # noqa: E402
from xnrs_amd.models.components import layers, news_encoding  # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
AG.LIVE_ROWS_MIN = 1
bad = []
for it in range(n_cfg):
    rng = np.random.default_rng(9000 + seed0 + it)
    h = int(rng.choice([1, 2, 4, 8]))
    dk = int(rng.choice([4, 8, 12, 16, 20, 32, 48, 64]))
    D = h * dk
    S = int(rng.choice([3, 7, 16, 20, 30, 32, 33, 40, 50, 64, 65, 80]))
    A = int(rng.choice([8, 33, 100, 256]))
    E = int(rng.choice([16, 32]))
    n_tab = int(rng.integers(8, 40))
    with_att = bool(rng.integers(0, 2)) or it % 3 == 0
    with_ids = bool(rng.integers(0, 2))
    want_dx = (not with_ids) and bool(rng.integers(0, 2))
    att = layers.MultiHeadAttention(h, D) if with_att else None
    enc = news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=E, in_features=D, att=att)
    enc.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, 100 + it))
    enc = enc.to(dev).eval()
    x = torch.from_numpy(rng.standard_normal((n_tab, S, D)).astype("float32"))
    m = torch.from_numpy((rng.random((n_tab, S)) < rng.uniform(0.2, 0.9)).astype("float32"))
    m[rng.random(n_tab) < 0.35] = 0
    m[0] = 0
    m[1, 0] = 1
    n_out = int(rng.integers(n_tab, 2 * n_tab)) if with_ids else n_tab
    ids = torch.from_numpy(rng.integers(0, n_tab, size=(n_out,)).astype("int64")) if with_ids else None
    w = torch.from_numpy(rng.standard_normal((n_out, E)).astype("float32")).to(dev)

    def run(live, kv):
        AG.LIVE_ROWS, AG.KV_ROWS = live, kv
        # the caching allocator hands freed blocks back: fill them with NaN first, so that any read of memory the row-list
        # paths leave unwritten (K|V / dQ|dK|dV rows of empty news) would poison the result
        junk = [torch.full((8 << 20,), float("nan"), device=dev) for _ in range(6)]
        del junk
        enc.zero_grad(set_to_none=True)
        xd = x.to(dev).requires_grad_(want_dx)
        if with_ids:
            y = enc.forward_ids(xd, m.to(dev), ids.to(dev).reshape(1, -1))[0][0]
        else:
            y = enc((xd.unsqueeze(0), m.to(dev).reshape(1, n_tab, S, 1)))[0][0]
        (y * w).sum().backward()
        return y.detach(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}, xd.grad

    tag = f"#{it} S={S} D={D} h={h} A={A} n={n_tab} att={with_att} ids={with_ids} dx={want_dx}"
    try:
        y0, g0, dx0 = run(False, False)
        for kv in (True, False):
            y1, g1, dx1 = run(True, kv)
            assert torch.equal(y0, y1), "forward differs"
            assert all(torch.isfinite(v).all() for v in g1.values()), "non-finite gradient"
            gmax = max(v.abs().max().item() for v in g0.values())
            for k in g0:
                sc = max(g0[k].abs().max().item(), 1e-3 * gmax)
                e = (g1[k] - g0[k]).abs().max().item() / sc
                assert e <= 2e-5, f"{k}: {e:.2e} (kv={kv})"
            if want_dx:
                sc = dx0.abs().max().item()
                assert (dx1 - dx0).abs().max().item() / sc <= 2e-5, "dx"
        print("ok  ", tag, flush=True)
    except Exception as e:  # noqa: BLE001
        bad.append((tag, repr(e)[:200]))
        print("FAIL", tag, repr(e)[:200], flush=True)
    finally:
        AG.LIVE_ROWS = AG.KV_ROWS = True
print(f"{n_cfg} configurations, {len(bad)} failures")
sys.exit(1 if bad else 0)
