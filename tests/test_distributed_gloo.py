"""CPU, world_size 2 and 4 (ragged shards), gloo: the N>1 path (impression sharding, ONE differentiable all-gather of
[user embedding | label bits] for the in-batch InfoNCE, ONE flat gradient all-reduce in a persistent bucket, no host sync) reproduces the single-process loss and gradients of the reference's
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


class _Count:
    """Counts the collectives issued between enter and exit (wraps the torch.distributed entry points the package uses)
    and fails on a host read of a tensor inside the step (`.item()` / `.tolist()`)."""
    NAMES = ("all_gather", "all_gather_into_tensor", "all_reduce", "broadcast", "reduce_scatter_tensor", "all_to_all_single")

    def __enter__(self):
        self.n = {k: 0 for k in self.NAMES}
        self.saved = {k: getattr(dist, k) for k in self.NAMES}
        for k in self.NAMES:
            def wrap(*a, _k=k, **kw):
                self.n[_k] += 1
                return self.saved[_k](*a, **kw)
            setattr(dist, k, wrap)
            setattr(D.dist, k, wrap)
        self.item, self.tolist = torch.Tensor.item, torch.Tensor.tolist
        self.syncs = 0

        def item(t):
            self.syncs += 1
            return self.item(t)

        def tolist(t):
            self.syncs += 1
            return self.tolist(t)
        torch.Tensor.item, torch.Tensor.tolist = item, tolist
        return self

    def __exit__(self, *exc):
        for k, f in self.saved.items():
            setattr(dist, k, f)
            setattr(D.dist, k, f)
        torch.Tensor.item, torch.Tensor.tolist = self.item, self.tolist
        return False

    @property
    def total(self):
        return sum(self.n.values())


def _worker(rank, world, port, out, overlapped=False):
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
    # ---- set-up, once per run: shard sizes (a collective + a host read, deliberately OUTSIDE the step) and the bucket
    layout = D.ShardLayout.exchange(hi - lo)
    assert layout.sizes == [b - a for a, b in (D.shard_range(C["B"], r, world) for r in range(world))]
    assert layout.equal == (C["B"] % world == 0)
    if overlapped:  # two buckets in the order their gradients complete: [user tower, news tower], reduced as they complete
        user = [v for k, v in sd.items() if not k.startswith("news_encoder.")]
        news = [v for k, v in sd.items() if k.startswith("news_encoder.")]
        bucket = D.OverlappedGradBuckets([user, news])
        assert len(bucket.buckets) == 2
    else:
        bucket = D.GradBucket(sd.values())
    n_reduce = 2 if overlapped else 1
    # ---- the step: exactly two collectives (three with the two overlapped buckets), no host read
    for _ in range(2):  # twice: the second pass shows the bucket's views survive a step (gradients do not accumulate)
        with _Count() as cnt:
            bucket.zero_grad()
            r = O.parent_forward(hist, cand, sd, C["h"])
            loss_rec = O.mse_relu_loss(r, batch["targets"])
            ue = O.parent_user_embeddings(hist, sd, C["h"])
            ue_all, lab_all = D.gather_embeddings_and_labels(ue, labels_full[lo:hi], layout)
            loss_cl = O.contrastive_loss(ue_all, lab_all, C["temperature"])
            loss = D.global_train_loss(loss_rec, hi - lo, C["B"], loss_cl, C["lambda_cl"])
            loss.backward()
            bucket.allreduce()
        assert cnt.total == 1 + n_reduce and cnt.n["all_gather_into_tensor"] == 1 and cnt.n["all_reduce"] == n_reduce, cnt.n
        assert cnt.syncs == 0, f"{cnt.syncs} host reads inside the step"
        assert torch.equal(lab_all, labels_full) and lab_all.dtype == labels_full.dtype
        for b in (bucket.buckets if overlapped else [bucket]):
            assert all(p.grad.data_ptr() >= b.flat.data_ptr() for p in b.params)
        if overlapped:  # the user tower's bucket completed by its hooks DURING the backward (before finish() was called)
            assert bucket._launched == [True, True]
    # the convenience forms (sizes exchanged inside) agree with the fused gather
    assert torch.equal(D.all_gather_rows(ue.detach()), ue_all.detach())
    assert torch.equal(D.all_gather_labels(labels_full[lo:hi]), labels_full)
    # the global loss value = sum over ranks of the weighted rec terms + lambda * cl
    rec = loss_rec.detach() * (hi - lo) / C["B"]
    dist.all_reduce(rec)
    if rank == 0:
        out["loss"] = (rec + C["lambda_cl"] * loss_cl.detach())
        out["grads"] = {k: v.grad.clone() for k, v in sd.items() if v.requires_grad}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,overlapped", [(2, False), (4, False), (2, True)])  # 6 impressions: 3+3 (equal shards) and 2+2+1+1 (ragged)
def test_n_rank_train_step_equals_single_process(world, overlapped):
    """overlapped: xnrs_amd.distributed.OverlappedGradBuckets -- one bucket per tower, all-reduced asynchronously as soon as
    the tower's last gradient has been accumulated (post-accumulate hooks), the rest of the backward running meanwhile."""
    loss1, g1 = _single()
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out, overlapped), nprocs=world, join=True)
    assert abs(out["loss"].item() - loss1.item()) <= 1e-6 * max(1.0, abs(loss1.item()))
    gmax = max(v.abs().max().item() for v in g1.values())
    for k, ref in g1.items():
        got = out["grads"][k]
        scale = max(ref.abs().max().item(), 1e-3 * gmax)
        assert (got - ref).abs().max().item() / scale <= 1e-4, k


def test_grad_bucket_reattaches_after_set_to_none():
    """optimizer.zero_grad(set_to_none=True) drops the views; the bucket notices and re-attaches (no stale gradients,
    no silent skip of the all-reduce)."""
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    b = D.GradBucket(ps)
    assert b.flat.numel() == 17 and b._attached()
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    assert torch.equal(b.flat, torch.cat([torch.full((12,), 2.0), torch.full((5,), 3.0)]))
    torch.optim.SGD(ps, lr=0.1).zero_grad(set_to_none=True)
    assert not b._attached()
    b.zero_grad()
    assert b._attached() and b.flat.abs().sum() == 0
    (ps[0].sum() + ps[1].sum()).backward()
    assert torch.equal(b.flat, torch.ones(17))


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 512):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _eval_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xnrs_amd.evaluation import _dist_rank_world, sharded_mean
    r, w, on = _dist_rank_world(None)
    assert (r, w, on) == (rank, world, True)
    per = torch.arange(37 * 9, dtype=torch.float64).reshape(37, 9).sin()  # per-impression metric rows of a 37-session epoch
    lo, hi = D.shard_range(37, rank, world)
    mean = sharded_mean(per[lo:hi].sum(0), 37, on)
    out[rank] = mean
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rank_sharded_metric_mean(world):
    """The reduction of the rank-sharded evaluation epoch (xnrs_amd.evaluation.evaluate): contiguous session blocks per rank
    (uneven: 37 sessions), ONE fp64 all-reduce of the per-rank sums, every rank ends with the single-process mean.  (The
    kernels of the epoch need a GPU: tests/test_hip_two_ranks.py runs the whole evaluate() with 2 and 4 ranks.)"""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_eval_worker, args=(world, port, out), nprocs=world, join=True)
        per = torch.arange(37 * 9, dtype=torch.float64).reshape(37, 9).sin()
        ref = per.sum(0) / 37
        for r in range(world):
            assert torch.allclose(out[r], ref, rtol=1e-13, atol=1e-15), r
