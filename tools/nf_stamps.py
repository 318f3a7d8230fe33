#!/usr/bin/env python
"""Where the fused news encoder (news_fused.hip) spends its cycles: per-wave s_memtime stamps at the phase boundaries
from the DIAGNOSTIC build (make -C xnrs_amd/csrc stamps -> libxnrs_hip_stamps.so), median over workgroups and waves.

    python tools/nf_stamps.py [S D h]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import hip  # noqa: E402

hip.LIB_PATH = os.path.join(ROOT, "xnrs_amd", os.environ.get("NF_LIB", "libxnrs_hip_stamps.so"))  # before the first hip.lib()
import bench  # noqa: E402
from xnrs_amd import synth  # noqa: E402

S, D, h = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (30, 320, 16)
n_news = 1024
dev = torch.device("cuda", 0)
w = dict(B=1, H=1, C=1, S=S, D=D, h=h, E=256 if D % 15 else 240, A=256)
model, _ = bench.build_model(w, dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
x, m = synth.device_tokens(gen, n_news, S, D, dev)
x, m = x.reshape(1, n_news, S, D), m.reshape(1, n_news, S, 1)
NST = 32
buf = torch.zeros(1024 * 8 * NST, dtype=torch.int64, device=dev)
lib = hip.lib()
lib.xnrs_debug_nf_set_stamps.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(20):  # warm clocks, stamps off
        model.news_encoder((x, m))
    torch.cuda.synchronize()
    assert lib.xnrs_debug_nf_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    model.news_encoder((x, m))
    torch.cuda.synchronize()
    lib.xnrs_debug_nf_set_stamps(None)
t = buf.cpu().numpy().reshape(1024, 8, NST)[: n_news // 2]
names = {0: "x rows issued", 1: "x in LDS (barrier)", 26: "barrier before Y store", 27: "Y in LDS (barrier)", 28: "fc1 k loop",
         29: "tanh + fc2", 30: "barrier", 31: "pool + store"}
for g in range(4):
    for i, nme in enumerate(("qkv k loop", "barrier", "qkv epilogue + barrier", "attention core", "barrier", "out-proj k loop")):
        names[2 + 6 * g + i] = f"g{g} {nme}"
d = np.diff(t, axis=2).astype(np.float64)  # [wg, wave, 31]
tot = (t[:, :, 31] - t[:, :, 0]).astype(np.float64)
print(f"S={S} D={D} h={h}: workgroup lifetime (stamp 0 -> 31) median {np.median(tot):.0f} cycles, "
      f"p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
groups = {}
for i in range(31):
    med = np.median(d[:, :, i])
    print(f"  {i + 1:2d} {names.get(i + 1, '?'):32s} median {med:9.0f}  mean {d[:, :, i].mean():9.0f}  max-wave median {np.median(d[:, :, i].max(axis=1)):9.0f}")
    key = names.get(i + 1, "?").split(" ", 1)[1] if names.get(i + 1, "?").startswith("g") else names.get(i + 1, "?")
    groups[key] = groups.get(key, 0.0) + d[:, :, i].mean()
for i, nme in ((7, "g1 qkv k loop"), (8, "g1 barrier after it"), (12, "g1 out-proj k loop")):
    print(f"  per-wave median of '{nme}':", " ".join(f"w{w}={np.median(d[:, w, i]):.0f}" for w in range(8)))
print("  -- summed over the head groups (mean cycles per wave)")
for k, v in groups.items():
    print(f"     {k:32s} {v:9.0f}  {100 * v / tot.mean():5.1f} %")
# launch span: first start to last end
print(f"  launch span {int(t[:, :, 31].max() - t[:, :, 0].min())} cycles; second-round workgroups start at "
      f"{np.median(np.sort(t[:, 0, 0] - t[:, :, 0].min())[256:]):.0f}")
