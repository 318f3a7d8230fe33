"""GPU, TWO processes sharing the one MI355X of the box: the data-parallel grad step with the HIP modules on both ranks
(xnrs_amd.distributed: one [embedding | label bits] all-gather, one flat gradient all-reduce, no host read) against the
single-process step over the whole batch.  RCCL refuses two ranks on one device, so the collectives ride on gloo, which
carries device tensors (tools/probes/gloo_cuda_probe.py) -- what is under test is the N>1 PATH with real HIP kernels on
every rank: sharding, the differentiable gather feeding the fused InfoNCE, the bucketed gradients, and the 1/N_global
loss weighting.  (tests/test_distributed_gloo.py: the same step on CPU with the oracle as the per-rank model, world 2 and
4; tests/test_hip_training.py: one rank over RCCL.)"""
import os
import socket
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

C = dict(model="NRMS", E=32, bias=False, h=4, D=32, H=8, S=6)
N_SESS, SEED = 48, 3


class Cfg(dict):
    __getattr__ = dict.__getitem__


def _world():
    from xnrs_amd import synth
    return synth.click_world(n_news=120, n_sess=64)


def _model(dev):
    from tests.golden import cases
    from xnrs_amd.models import make_model
    cfg = Cfg(cases.model_cfg(C))
    cfg["p_dropout"] = 0.0
    torch.manual_seed(0)
    return make_model(cfg).to(dev).eval()


def _batch(dev, lo, hi):
    from xnrs_amd.data import DeviceBatcher
    store, beh = _world()
    store, beh = store.to(dev), beh.to(dev)
    sess = torch.arange(N_SESS, device=dev)
    hist, cand, targets = DeviceBatcher(beh, l_hist=8).train_batch(sess, n_neg=4, seed=SEED)
    labels = beh.theme_labels[sess]
    return store, hist[lo:hi], cand[lo:hi], targets[lo:hi], labels[lo:hi]


def _rank(rank, world, port, path):
    import torch.distributed as dist

    from xnrs_amd import distributed as D
    from xnrs_amd.losses import contrastive_loss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    lo, hi = D.shard_range(N_SESS, rank, world)
    store, hist, cand, targets, labels = _batch(dev, lo, hi)
    model = _model(dev)
    D.broadcast_parameters(model)
    layout = D.ShardLayout.uniform(hi - lo)
    bucket = D.GradBucket(model.parameters())
    for _ in range(2):  # the second pass: the bucket's views survive a step
        bucket.zero_grad()
        r, u, _ = model.forward_ids(store.x, store.m, hist, cand, return_embeddings=True)
        rec = torch.nn.functional.mse_loss(torch.relu(r), targets)
        ue, lab = D.gather_embeddings_and_labels(u.squeeze(1), labels, layout)
        loss = D.global_train_loss(rec, hi - lo, N_SESS, contrastive_loss(ue, lab, 0.08), 0.1)
        loss.backward()
        bucket.allreduce()
    assert bucket._attached() and ue.is_cuda and lab.shape[0] == N_SESS
    part = rec.detach() * (hi - lo) / N_SESS
    dist.all_reduce(part)
    if rank == 0:
        torch.save({"loss": (part + 0.1 * contrastive_loss(ue.detach(), lab, 0.08)).cpu(),
                    "grads": {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}}, path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_equal_the_single_process_step():
    import torch.multiprocessing as mp

    from xnrs_amd.losses import contrastive_loss
    dev = torch.device("cuda", 0)
    store, hist, cand, targets, labels = _batch(dev, 0, N_SESS)
    model = _model(dev)
    r, u, _ = model.forward_ids(store.x, store.m, hist, cand, return_embeddings=True)
    loss = torch.nn.functional.mse_loss(torch.relu(r), targets) + 0.1 * contrastive_loss(u.squeeze(1), labels, 0.08)
    loss.backward()
    ref = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "rank0.pt")
        mp.spawn(_rank, args=(2, port, path), nprocs=2, join=True)  # 2 GPU processes + this one: within the box's limit
        got = torch.load(path, weights_only=True)
    assert abs(got["loss"].item() - loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))
    # (the bucket gives EVERY parameter a gradient view: the ones the single-process step leaves at None -- dummy_param --
    # must be exact zeros)
    assert set(ref) <= set(got["grads"]) and len(ref) >= 20
    for k in set(got["grads"]) - set(ref):
        assert not got["grads"][k].any(), k
    gmax = max(v.abs().max().item() for v in ref.values())
    for k, g in ref.items():
        scale = max(g.abs().max().item(), 1e-3 * gmax)
        assert (got["grads"][k] - g).abs().max().item() / scale <= 1e-4, k


def _eval_rank(rank, world, port, path):
    import torch.distributed as dist

    from xnrs_amd import distributed as D
    from xnrs_amd.evaluation import evaluate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    store, beh = _world()
    store, beh = store.to(dev), beh.to(dev)
    model = _model(dev)
    D.broadcast_parameters(model)
    res = evaluate(model, store, beh, l_hist=8, batch=16)  # several batches per rank, a ragged last one
    torch.save(res, f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_rank_sharded_evaluation_equals_the_single_process_epoch(world):
    """evaluate() under a process group: news table encoded in slices + one all-gather, sessions split by shard_range, one
    fp64 all-reduce of the metric sums (the reference's test loop, training.py:194-243, is one process at batch size 1) --
    every rank returns the single-process metrics (uneven shards: 121 table rows, 64 sessions over 4 ranks of batch 16)."""
    import torch.multiprocessing as mp

    from xnrs_amd.evaluation import evaluate
    dev = torch.device("cuda", 0)
    store, beh = _world()
    store, beh = store.to(dev), beh.to(dev)
    ref = evaluate(_model(dev), store, beh, l_hist=8, batch=16, distributed=False)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "res")
        mp.spawn(_eval_rank, args=(world, port, path), nprocs=world, join=True)  # <= 4 GPU processes + this one
        got = [torch.load(f"{path}.{r}", weights_only=True) for r in range(world)]
    for res in got:
        assert res.keys() == ref.keys()
        for k in ref:
            assert abs(res[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (k, res[k], ref[k])
    assert all(g == got[0] for g in got)  # every rank holds the same dict
