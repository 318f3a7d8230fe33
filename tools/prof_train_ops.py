#!/usr/bin/env python
"""Which torch operators of ONE grad step launch device kernels, and from where: torch.profiler over one step of
bench.make_train_job (NRMS by default), printed as (operator, kernel launches) with the Python frame that called it.
    python3 tools/prof_train_ops.py [nrms|standard|naml]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "nrms"
model, opt, batch, targets, labels, fn = bench.make_train_job(name, dev)
for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    fn()
    torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("device kernels in the step:", len(kern))
by = collections.Counter()
where = {}
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    # innermost operator that owns the kernels
    if any(c.kernels for c in e.cpu_children):
        continue
    st = [s for s in (e.stack or []) if "/root/repo" in s or "xnrs_amd" in s or "bench.py" in s]
    key = (e.name, st[0].split("/")[-1] if st else "-")
    by[key] += len(e.kernels)
for (n, s), c in by.most_common(60):
    print(f"{c:4d}  {n:45s} {s}")
names = collections.Counter(k.name[:70] for k in kern)
print("--- kernels by name")
for n, c in names.most_common(80):
    print(f"{c:4d}  {n}")

print("--- torch (non-xnrs) kernels by device time")
tk = sorted([k for k in kern if "xnrs::" not in k.name], key=lambda k: -k.device_time_total if hasattr(k, "device_time_total") else 0)
for k in tk[:12]:
    print(f"{getattr(k, 'device_time_total', 0):8.1f} us  {k.name[:100]}")
print("--- operators owning them (name, input shapes, device time of the op's kernels)")
rows = []
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels or any(c.kernels for c in e.cpu_children):
        continue
    if any("xnrs::" in k.name for k in e.kernels):
        continue
    rows.append((sum(k.duration for k in e.kernels), e.name, str(e.input_shapes)[:120], [s for s in (e.stack or []) if "repo" in s][:2]))
for r in sorted(rows, key=lambda r: -r[0])[:12]:
    print(f"{r[0]:8.1f} us  {r[1]:28s} {r[2]}  {r[3]}")
