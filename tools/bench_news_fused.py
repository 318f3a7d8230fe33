#!/usr/bin/env python
"""BASELINE configs[1]: NRMS news encoder only, 1024 news x 30 tokens -- the fused short-title kernel
(news_fused.hip) against the six-launch pipeline (XNRS_NEWS_FUSED=0), interleaved rounds in one process."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import hip, synth  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [(1024, 30, 300, 15), (1024, 30, 320, 16), (1024, 20, 320, 16), (1024, 32, 320, 16)]
if os.environ.get("NF_SWEEP"):  # dispatch-rule sweep: news count and title length
    shapes = [(n, 30, 320, 16) for n in (64, 256, 512, 2048, 8192, 28160)] + [(1024, S, 320, 16) for S in (16, 22, 24, 26, 28)]
for (n_news, S, D, h) in shapes:
    w = dict(B=1, H=1, C=1, S=S, D=D, h=h, E=256 if D != 300 else 240, A=256)
    model, _ = bench.build_model(w, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x, m = synth.device_tokens(gen, n_news, S, D, dev)
    x, m = x.reshape(1, n_news, S, D), m.reshape(1, n_news, S, 1)
    fl = n_news * bench.news_flops(S, D, 256, w["E"], folded=True)  # executed (out-projection folded behind the pooling)
    fl_ref = n_news * bench.news_flops(S, D, 256, w["E"])           # at the reference's operation order
    res = {"1": [], "0": [], "npw1": [], "nofold": []}
    with torch.no_grad():
        for rnd in range(5):
            for flag in ("1", "0", "npw1", "nofold"):
                with hip.knobs(XNRS_NEWS_FUSED="2" if flag != "0" else "0", XNRS_NEWS_FUSED_NPW="1" if flag == "npw1" else "2",
                               XNRS_FOLD_OUT="0" if flag == "nofold" else "1"):
                    for _ in range(5):
                        model.news_encoder((x, m))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        model.news_encoder((x, m))
                    torch.cuda.synchronize()
                    res[flag].append((time.perf_counter() - t0) / 20)
        with hip.knobs(XNRS_NEWS_FUSED="2"):
            hip.profile_enable(hip.PROFILE_ALL)
            for _ in range(10):
                model.news_encoder((x, m))
            torch.cuda.synchronize()
            st = hip.profile_read()
            hip.profile_enable(0)
    f, u, f1, nf = sorted(res["1"])[2], sorted(res["0"])[2], sorted(res["npw1"])[2], sorted(res["nofold"])[2]
    kms = st["news_fused"][0] / max(st["news_fused"][1], 1)
    print(f"S={S} D={D} h={h} n={n_news}: fused {f*1e6:.1f} us ({fl/f/1e12:.1f} TF executed = {fl/f/1e12/157.3:.3f} of fp32 MFMA peak; "
          f"{fl_ref/f/1e12:.1f} TF at the reference's operation order; kernel alone {kms*1e3:.1f} us)  1 news/WG {f1*1e6:.1f} us  "
          f"per-token out-projection in the kernel {nf*1e6:.1f} us ({fl_ref/nf/1e12:.1f} TF)  pipeline {u*1e6:.1f} us ({fl/u/1e12:.1f} TF)", flush=True)
