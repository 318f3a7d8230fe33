#!/usr/bin/env python
"""End-to-end example in the shape of the reference's train.py / ContrastiveRankingTrainer (training.py:402-431),
entirely on the device: resident news table -> device batch assembly -> forward_ids (id gather fused into the first
GEMM) -> relu/MSE + lambda * fused InfoNCE -> hand-written HIP backward -> Adam -> device evaluation.

    python examples/train_synthetic.py [--model NRMS|standard] [--steps 300] [--batch 64] [--bf16x3]

Synthetic click world (xnrs_amd.synth.click_world): news carry a topic direction in their tokens, a user clicks news of
its own topic -- so the ranking metrics have to rise if the whole stack (forward, gradients, optimiser, evaluation) is
right.  Needs an MI355X; there is no CPU fallback."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xnrs_amd import evaluation, hip, synth  # noqa: E402
from xnrs_amd.data import DeviceBatcher  # noqa: E402
from xnrs_amd.losses import contrastive_loss  # noqa: E402
from xnrs_amd.models import make_model  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.__getitem__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="NRMS", choices=("NRMS", "standard"))
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hist", type=int, default=8)
    ap.add_argument("--bf16x3", action="store_true", help="forward GEMMs + dX products on the bf16 matrix cores (fp32-grade)")
    args = ap.parse_args()
    dev = "cuda:0"
    if args.bf16x3:
        hip.set_gemm_mode(hip.GEMM_BF16X3)
    torch.manual_seed(0)
    store, beh = synth.click_world(n_news=2000, n_sess=4000, S=12, D=64, n_topics=6)
    store, beh = store.to(dev), beh.to(dev)
    cfg = Cfg(synth.model_cfg(dict(model=args.model, E=64, bias=True, h=4, D=64, H=args.hist, S=12)))
    model = make_model(cfg).to(dev)
    model.news_encoder.skip_empty = True          # empty history slots share one encoded representative (exact)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    batcher = DeviceBatcher(beh, l_hist=args.hist)
    model.eval()
    before = evaluation.evaluate(model, store, beh, args.hist, batch=1024)
    print("before:", {k: round(v, 4) for k, v in before.items()})
    model.train()
    t0 = time.perf_counter()
    for step in range(args.steps):
        sess = torch.randint(0, len(beh), (args.batch,), device=dev)
        hist, cand, targets = batcher.train_batch(sess, n_neg=4, seed=step)
        opt.zero_grad()
        r, u, _ = model.forward_ids(store.x, store.m, hist, cand, return_embeddings=True)
        loss = torch.nn.functional.mse_loss(torch.relu(r), targets) + \
            0.1 * contrastive_loss(u.squeeze(1), beh.theme_labels[sess], 0.08)
        loss.backward()
        opt.step()
        if step % 50 == 0 or step == args.steps - 1:
            print(f"step {step:4d}  loss {loss.item():.4f}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    after = evaluation.evaluate(model, store, beh, args.hist, batch=1024)
    print("after: ", {k: round(v, 4) for k, v in after.items()})
    print(f"{args.steps} steps of {args.batch} impressions in {dt:.2f} s = {args.steps * args.batch / dt:.0f} impressions/s (tiny model: launch-bound)")
    assert after["auc"] > before["auc"] + 0.1, "the model did not learn"


if __name__ == "__main__":
    main()
