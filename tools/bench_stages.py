#!/usr/bin/env python
"""Per-stage time of the NRMS news encoder under development knobs (env vars), interleaved rounds in
one process.  usage: python tools/bench_stages.py "NAME=ENV1=v,ENV2=v;NAME2=..." [n_news S D h]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xnrs_amd import hip, synth  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "base="
variants = {}
for item in spec.split(";"):
    name, _, envs = item.partition("=")
    variants[name] = dict(e.split("=") for e in envs.split(",") if e)
n_news, S, D, h = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1310 * 4, 50, 768, 16)
dev = torch.device("cuda", 0)
w = dict(B=1, H=1, C=1, S=S, D=D, h=h, E=256 if D % 15 else 240, A=256)
model, _ = bench.build_model(w, dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
x, m = synth.device_tokens(gen, n_news, S, D, dev)
x, m = x.reshape(1, n_news, S, D), m.reshape(1, n_news, S, 1)
all_keys = set(k for v in variants.values() for k in v)
res = {v: [] for v in variants}
with torch.no_grad():
    for rnd in range(4):
        for v, env in variants.items():
            for k in all_keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            hip.reload_knobs()  # the library reads its knobs at load and on request only
            for _ in range(2):
                model.news_encoder((x, m))
            torch.cuda.synchronize()
            hip.profile_enable(hip.PROFILE_ALL)
            for _ in range(3):
                model.news_encoder((x, m))
            torch.cuda.synchronize()
            st = hip.profile_read()
            hip.profile_enable(0)
            res[v].append({k: t[0] / 3 for k, t in st.items()})
for v, rs in res.items():
    med = {k: sorted(r[k] for r in rs)[len(rs) // 2] for k in rs[0]}
    print(v, " ".join(f"{k}={t:.3f}ms" for k, t in med.items()), f"total={sum(med.values()):.3f}ms")
