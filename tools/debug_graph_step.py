#!/usr/bin/env python
"""Which stage of the grad step captured in a hipGraph differs from the eager step, per configuration (the Q|K|V cache is
cleared before the capture, so the captured step cannot read an image of the warm-up step).  Two findings came out of it
(round 4; xnrs_amd/csrc/kernels.h load_dev_scalar, tests/test_hip_train_step.py):
  * CAPTURE_ON_SIDE=0 (torch's own capture stream instead of the warm-up stream): replays are wrong -- gradient accumulation
    is captured as a forked branch (AccumulateGrad stream mismatch) and block reuse in the graph pool corrupts it;
  * without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment (ROCm 7.2, gfx950): later replays read stale data (NaN);
    with it 128 of 128 replays equal the eager step bit for bit."""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden import cases  # noqa: E402
from tests.test_hip_grads import Cfg, load  # noqa: E402
from xnrs_amd import autograd as AG, synth  # noqa: E402
from xnrs_amd.losses import contrastive_loss  # noqa: E402
from xnrs_amd.models import make_model  # noqa: E402

DEV = "cuda:0"
AG.LIVE_ROWS_MIN = 1
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
configs = [dict(live=True, kv=True, share=True, merge=True), dict(live=True, kv=True, share=True, merge=False),
           dict(live=True, kv=False, share=True, merge=False), dict(live=True, kv=True, share=False, merge=False)]
for cf in configs:
    AG.LIVE_ROWS, AG.KV_ROWS, AG.SHARE_QKV, AG.MERGE_DW = cf["live"], cf["kv"], cf["share"], cf["merge"]
    report = []
    for trial in range(TRIALS):
        c = dict(model="NRMS", B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=False, seed=4303 + trial, min_len=3)
        model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
        model.train()
        batch = synth.batch_to(cases.model_batch(c), DEV)
        hist = batch["user_features"]["history"]["title_emb"]
        cand = batch["candidate_features"]["title_emb"]
        labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
        named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        for _, p in named:
            p.grad = torch.zeros_like(p)

        def step():
            for _, p in named:
                p.grad.zero_()
            h, hm = model.news_encoder(hist)
            cv, _ = model.news_encoder(cand)
            u = model.user_encoder((h, hm))
            r = model.rec_model(u, cv)
            ue = model.get_user_embeddings(batch)
            loss = torch.nn.functional.mse_loss(torch.relu(r), batch["targets"]) + 0.1 * contrastive_loss(ue, labels, 0.08)
            keep = {"h": h.detach().clone(), "c": cv.detach().clone(), "u": u.detach().clone(), "ue": ue.detach().clone(),
                    "loss": loss.detach().clone()}
            loss.backward()
            return loss, keep

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            torch.manual_seed(77)
            _, k0 = step()
            g0 = [p.grad.clone() for _, p in named]
            torch.manual_seed(77)
            step()
        torch.cuda.current_stream().wait_stream(side)
        if os.environ.get("NOCLEAR", "0") != "1":
            gc.collect()
            AG._QKV_IMAGES.clear()
            AG._OUTPUTS.clear()
        graph = torch.cuda.CUDAGraph()
        torch.manual_seed(77)
        st0 = dict(AG.STATS)
        with torch.cuda.graph(graph, stream=side if os.environ.get('CAPTURE_ON_SIDE', '1') == '1' else None):
            lg, kg = step()
        st = {k: AG.STATS[k] - st0[k] for k in AG.STATS if AG.STATS[k] != st0[k]}
        for rep in range(2):
            graph.replay()
            torch.cuda.synchronize()
            bad = [k for k in k0 if not torch.equal(k0[k], kg[k])]
            badg = [(n, float((p.grad - g).abs().max() / (g.abs().max() + 1e-30))) for (n, p), g in zip(named, g0) if not torch.equal(p.grad, g)]
            worst = max(badg, key=lambda t: t[1]) if badg else None
            report.append((trial, rep, bad, len(badg), worst, st if rep == 0 else None))
        del graph
    print(f"{cf}:", [r for r in report if r[2] or r[3]] or "all replays equal", flush=True)
