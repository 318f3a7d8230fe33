"""CPU, world_size 2, gloo: the N>1 path (impression sharding, differentiable all-gather for the in-batch
InfoNCE, flat gradient all-reduce) reproduces the single-process loss and gradients of the reference's
train step.  The per-rank model here is the CPU oracle (the HIP modules need a GPU); what is under test
is xnrs_amd.distributed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases
from xnrs_amd import distributed as D

C = dict(model="NRMS", B=6, H=3, C=3, S=8, D=32, h=4, E=16, bias=False, seed=500, temperature=0.08, lambda_cl=0.1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single():
    sd = {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in H.state_for(H.model_shapes(C), C["seed"] + 1).items()}
    batch = cases.model_batch(C)
    labels = cases.theme_labels(batch["main_theme"])
    loss, _, _ = O.train_step_loss(batch, sd, C["h"], labels, C["temperature"], C["lambda_cl"])
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sd = {k: torch.nn.Parameter(v.clone(), requires_grad=not k.endswith("dummy_param"))
          for k, v in H.state_for(H.model_shapes(C), C["seed"] + 1).items()}
    full = cases.model_batch(C)
    labels_full = cases.theme_labels(full["main_theme"])
    lo, hi = D.shard_range(C["B"], rank, world)
    batch = D.shard_batch(full, rank, world)
    assert batch["targets"].shape[0] == hi - lo and len(batch["main_theme"]) == hi - lo
    hist = batch["user_features"]["history"]["title_emb"]
    cand = batch["candidate_features"]["title_emb"]
    r = O.parent_forward(hist, cand, sd, C["h"])
    loss_rec = O.mse_relu_loss(r, batch["targets"])
    ue = O.parent_user_embeddings(hist, sd, C["h"])
    ue_all = D.all_gather_rows(ue)
    lab_all = D.all_gather_labels(labels_full[lo:hi])
    assert torch.equal(lab_all, labels_full)
    loss_cl = O.contrastive_loss(ue_all, lab_all, C["temperature"])
    loss = D.global_train_loss(loss_rec, hi - lo, C["B"], loss_cl, C["lambda_cl"])
    loss.backward()
    D.allreduce_gradients(sd.values())
    # the global loss value = sum over ranks of the weighted rec terms + lambda * cl
    rec = loss_rec.detach() * (hi - lo) / C["B"]
    dist.all_reduce(rec)
    if rank == 0:
        out["loss"] = (rec + C["lambda_cl"] * loss_cl.detach())
        out["grads"] = {k: v.grad.clone() for k, v in sd.items() if v.requires_grad}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_train_step_equals_single_process():
    loss1, g1 = _single()
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert abs(out["loss"].item() - loss1.item()) <= 1e-6 * max(1.0, abs(loss1.item()))
    gmax = max(v.abs().max().item() for v in g1.values())
    for k, ref in g1.items():
        got = out["grads"][k]
        scale = max(ref.abs().max().item(), 1e-3 * gmax)
        assert (got - ref).abs().max().item() / scale <= 1e-4, k


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 512):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
