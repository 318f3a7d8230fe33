"""GPU end-to-end: the whole stack in the shape of the reference's training loop (training.py:402-431) --
device batch assembly -> forward_ids (id gather fused in the first GEMM) -> relu/MSE + lambda * fused InfoNCE ->
HIP backward -> Adam -> device evaluation -- learns a synthetic click model."""
import numpy as np
import pytest
import torch

from tests.golden import cases
from xnrs_amd import evaluation as EV
from xnrs_amd import synth
from xnrs_amd.data import Behaviors, DeviceBatcher, NewsStore
from xnrs_amd.losses import contrastive_loss
from xnrs_amd.models import make_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg(dict):
    __getattr__ = dict.__getitem__


synthetic_world = synth.click_world


@pytest.mark.parametrize("model_name", ["NRMS", "standard"])
def test_training_learns_and_eval_improves(model_name):
    torch.manual_seed(0)
    store, beh = synthetic_world()
    store, beh = store.to(DEV), beh.to(DEV)
    c = dict(model=model_name, E=32, bias=False, h=4, D=32, H=8, S=6)
    cfg = Cfg(cases.model_cfg(c))
    cfg["p_dropout"] = 0.0
    model = make_model(cfg).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    batcher = DeviceBatcher(beh, l_hist=8)
    model.eval()
    before = EV.evaluate(model, store, beh, 8, batch=128)
    model.train()  # attention dropout 0.1 active for NRMS, like the reference
    losses = []
    n = len(beh)
    for step in range(80):
        sess = torch.randint(0, n, (64,), device=DEV)
        hist, cand, targets = batcher.train_batch(sess, n_neg=4, seed=step)
        opt.zero_grad()
        r, u, _ = model.forward_ids(store.x, store.m, hist, cand, return_embeddings=True)
        loss = torch.nn.functional.mse_loss(torch.relu(r), targets) + 0.1 * contrastive_loss(u.squeeze(1), beh.theme_labels[sess], 0.08)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    assert np.mean(losses[-10:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    model.eval()
    after = EV.evaluate(model, store, beh, 8, batch=128)
    # ReLU'd scores tie at 0 for many negatives (AUC counts ties as 1/2), so the bar is relative
    assert after["auc"] > before["auc"] + 0.15, (before, after)
    assert after["ndcg@10"] > before["ndcg@10"] and after["ctr@1"] > before["ctr@1"] + 0.2, (before, after)


def test_one_rank_rccl_train_step_equals_plain_step():
    """The N>1 train step (xnrs_amd.distributed over backend "nccl" = RCCL) with HIP modules and device tensors.
    One GPU here, so world_size 1: every collective runs through RCCL on device memory, and the result must
    equal the plain single-process step bit for bit.  (World size 2 is covered on CPU/gloo by
    tests/test_distributed_gloo.py; two ranks cannot share one device under RCCL.)"""
    import os
    import socket

    import torch.distributed as dist

    from xnrs_amd import distributed as D

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    c = dict(model="NRMS", E=32, bias=False, h=4, D=32, H=8, S=6)
    cfg = Cfg(cases.model_cfg(c))
    cfg["p_dropout"] = 0.0
    store, beh = synthetic_world(n_news=120, n_sess=64)
    store, beh = store.to(DEV), beh.to(DEV)
    sess = torch.arange(48, device=DEV)
    hist, cand, targets = DeviceBatcher(beh, l_hist=8).train_batch(sess, n_neg=4, seed=3)
    labels = beh.theme_labels[sess]

    def step(distributed):
        torch.manual_seed(0)
        model = make_model(cfg).to(DEV).eval()
        if distributed:
            D.broadcast_parameters(model)
            bucket = D.GradBucket(model.parameters())  # .grad = views of one flat buffer
            bucket.zero_grad()
        r, u, _ = model.forward_ids(store.x, store.m, hist, cand, return_embeddings=True)
        rec = torch.nn.functional.mse_loss(torch.relu(r), targets)
        if distributed:  # the step's two collectives over RCCL: [embedding | label bits] all-gather, flat all-reduce
            ue, lab = D.gather_embeddings_and_labels(u.squeeze(1), labels, D.ShardLayout.uniform(r.shape[0]))
            assert torch.equal(lab, labels)
            loss = D.global_train_loss(rec, r.shape[0], r.shape[0], contrastive_loss(ue, lab, 0.08), 0.1)
        else:
            loss = rec + 0.1 * contrastive_loss(u.squeeze(1), labels, 0.08)
        loss.backward()
        if distributed:
            bucket.allreduce()
            assert bucket._attached()
        return loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    l0, g0 = step(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        l1, g1 = step(True)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    assert torch.equal(l0, l1) and len(g0) > 0
    for k, a in g0.items():
        assert torch.equal(a, g1[k]), k


def test_forward_and_backward_are_bitwise_reproducible():
    """No float atomics anywhere (split-K slabs, column sums, dQ partials and the InfoNCE reductions are all summed in
    a fixed order): two runs of the same step -- same inputs, same dropout seed -- give bit-identical scores and
    gradients at a size where every kernel spans many workgroups."""
    import bench
    dev = torch.device(DEV)
    w = dict(bench.WORKLOAD, B=48, H=20)
    model, _ = bench.build_model(w, dev)
    model.train()  # attention dropout on: the counter-based RNG must reproduce as well
    hist, cand = bench.make_inputs(w, dev, seed=21)
    targets = torch.zeros(w["B"], w["C"], 1, device=dev)
    targets[:, 0] = 1.0
    labels = torch.arange(w["B"], device=dev) % 5

    def run():
        torch.manual_seed(1234)
        model.zero_grad(set_to_none=True)
        r, u, _ = model._forward(hist, cand, return_embeddings=True)
        loss = torch.nn.functional.mse_loss(torch.relu(r), targets) + 0.1 * contrastive_loss(u.squeeze(1), labels, 0.08)
        loss.backward()
        return r.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    r0, g0 = run()
    r1, g1 = run()
    assert torch.equal(r0, r1)
    assert g0.keys() == g1.keys() and len(g0) > 20
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
