#!/usr/bin/env python
"""Soak: the device-compacted padding-free encoder (xnrs_text_encoder_fwd_compact) against the padded call over random
shapes, news counts spanning several passes (small chunk), prefix masks (bitwise equal) and masks with holes (<= 2e-6:
the pooling normaliser's summation order), +- attention tower, +- head, +- id table, the workspace and the allocator's free
blocks NaN-filled first.   python tools/soak_compact.py [n] [seed0]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip, ops, synth  # noqa: E402
from xnrs_amd.models.components import layers, news_encoding  # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
bad = []
for it in range(n_cfg):
    rng = np.random.default_rng(7000 + seed0 + it)
    h = int(rng.choice([1, 2, 4, 8]))
    dk = int(rng.choice([4, 8, 12, 16, 20, 32, 48, 64]))
    D = h * dk
    S = int(rng.choice([1, 3, 7, 16, 20, 30, 32, 33, 40, 50, 64]))
    A = int(rng.choice([8, 33, 100, 256]))
    E = int(rng.choice([16, 32]))
    n = int(rng.integers(1, 300))
    with_att = bool(rng.integers(0, 2)) or it % 3 == 0
    with_head = bool(rng.integers(0, 2))
    with_ids = bool(rng.integers(0, 2))
    holes = bool(rng.integers(0, 2))
    chunk = int(rng.choice([0, 7, 64]))
    att = layers.MultiHeadAttention(h, D) if with_att else None
    enc = news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=E if with_head else D,
                                    in_features=D, head=with_head, att=att)
    enc.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, 300 + it))
    enc = enc.to(dev).eval()
    n_tab = n if not with_ids else int(rng.integers(1, 60))
    x = torch.from_numpy(rng.standard_normal((n_tab, S, D)).astype("float32")).to(dev)
    if holes:
        m = (rng.random((n_tab, S)) < rng.uniform(0.2, 0.9)).astype("float32")
    else:
        m = (np.arange(S)[None, :] < rng.integers(0, S + 1, size=(n_tab, 1))).astype("float32")
    m[rng.random(n_tab) < 0.3] = 0
    m = torch.from_numpy(m).to(dev)
    ids = torch.from_numpy(rng.integers(0, n_tab, size=(n,)).astype("int32")).to(dev) if with_ids else None
    tag = f"#{it} S={S} D={D} h={h} A={A} n={n} att={with_att} head={with_head} ids={with_ids} holes={holes} chunk={chunk}"
    try:
        if not ops.compact_supported(S, D, att, enc.pooler):
            print("skip", tag, flush=True)
            continue
        with torch.no_grad():
            with hip.knobs(XNRS_NEWS_FUSED="0"):  # the padded GEMM pipeline (the fused short-title kernel differs by ~1e-7)
                y0, hm0 = ops.text_encoder(x, m.unsqueeze(-1), enc, ids=ids)
            hip.release_workspaces()
            junk = [torch.full((8 << 20,), float("nan"), device=dev) for _ in range(6)]
            del junk
            y1, hm1 = ops.text_encoder_forward_compact(x, m.unsqueeze(-1), att, enc.pooler, getattr(enc, "head", None), ids=ids,
                                                       chunk=chunk)
        assert torch.isfinite(y1).all(), "non-finite"
        assert torch.equal(hm0.reshape(-1), hm1.reshape(-1)), "news mask"
        if holes:
            sc = y0.abs().max().item()
            assert (y1 - y0).abs().max().item() <= 2e-6 * sc, f"{(y1 - y0).abs().max().item() / sc:.2e}"
        else:
            assert torch.equal(y0, y1), f"not bitwise: {(y1 - y0).abs().max().item():.2e}"
        print("ok  ", tag, flush=True)
    except Exception as e:  # noqa: BLE001
        bad.append((tag, repr(e)[:200]))
        print("FAIL", tag, repr(e)[:200], flush=True)
print(f"{n_cfg} configurations, {len(bad)} failures")
sys.exit(1 if bad else 0)
