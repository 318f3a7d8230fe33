#!/usr/bin/env python
"""Soak: the benchmark step N times, every result bitwise equal to the first (races in the hand-written kernels show up as
run-to-run differences); then the NRMS grad step N times from the same weights, every gradient bitwise equal."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
model, _ = bench.build_model(w, dev)
hist, cand = bench.make_inputs(w, dev, seed=123)
bad = 0
with torch.no_grad():
    ref = bench.step(model, hist, cand).clone()
    for i in range(N):
        bad += int(not torch.equal(bench.step(model, hist, cand), ref))
print(f"forward: {N} steps, {bad} differ from the first", flush=True)
with torch.no_grad():  # the device-compacted padding-free encoder: equal to the dense step, every time
    model.news_encoder.unpadded = True
    for i in range(N):
        bad += int(not torch.equal(bench.step(model, hist, cand), ref))
    model.news_encoder.unpadded = False
print(f"forward, device-compacted: {N} steps, {bad} differ from the dense first step (cumulative)", flush=True)

w2 = dict(B=64, H=25, C=5, S=50, D=768, h=16, E=256, A=256)
m2, _ = bench.build_model(w2, dev)
m2.train()  # attention dropout 0.1 ON, the same seeds every step (torch.manual_seed below): identical steps through the
            # shared projection AND the merged weight-gradient products (eval mode would share the second encode whole)
h2, c2 = bench.make_inputs(w2, dev, seed=7)
batch = {"user_features": {"history": {"title_emb": h2}, "other": {}}, "candidate_features": {"title_emb": c2}}
tgt = torch.zeros(64, 5, 1, device=dev)
tgt[:, 0] = 1


def grads():
    # the reference's step: scores, then the user embeddings of a second history encode (which reads the first one's Q|K|V
    # image, autograd._QKV_IMAGES) feeding a second loss term
    m2.zero_grad(set_to_none=True)
    torch.manual_seed(5)
    loss = torch.nn.functional.mse_loss(torch.relu(m2(batch)), tgt)
    loss = loss + 0.1 * m2.get_user_embeddings(batch).square().mean()
    loss.backward()
    return [p.grad.clone() for p in m2.parameters() if p.grad is not None]


g0 = grads()
badg = 0
for i in range(N):
    g = grads()
    badg += int(any(not torch.equal(a, b) for a, b in zip(g, g0)))
print(f"grad step: {N} steps, {badg} differ from the first")
sys.exit(1 if bad or badg else 0)
