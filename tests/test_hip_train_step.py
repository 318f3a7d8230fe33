"""GPU: the grad step as a first-class path (round 4) -- row lists built on the device (no host read of the counts), ONE
weight-gradient product per projection for the reference's two history encodes (training.py:406,409), the whole step
captured in a hipGraph, and the sharing the bench relies on for StandardRec proven exact."""
import pytest
import torch

from tests import helpers as H
from tests.golden import cases
from tests.test_hip_grads import Cfg, load
from xnrs_amd import hip, synth
from xnrs_amd.losses import contrastive_loss
from xnrs_amd.models import make_model
from xnrs_amd.models.components import layers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,L,with_ids", [(1, 7, False), (70, 50, False), (333, 24, True), (130, 100, False), (2049, 5, True)])
def test_device_row_lists_equal_torch_nonzero(n, L, with_ids):
    """xnrs_build_row_lists against the torch bookkeeping it replaces (autograd._host_lists): live rows, the token rows of
    the non-empty sequences, their table rows with ids, and both counts."""
    import ctypes as C
    rng = synth.rng_for(4000 + n)
    n_tab = n + 17 if with_ids else n
    m = torch.from_numpy((rng.random((n_tab, L)) < 0.5).astype("float32"))
    m[torch.from_numpy(rng.random(n_tab) < 0.35)] = 0
    ids = torch.from_numpy(rng.integers(0, n_tab, size=(n,)).astype("int32")) if with_ids else None
    md = m.to(DEV)
    idd = ids.to(DEV) if with_ids else None
    cap = n * L
    buf = torch.full((4, cap), -7, dtype=torch.int32, device=DEV)
    counts = torch.zeros(2, dtype=torch.int64, device=DEV)
    l = hip.lib()
    nws = l.xnrs_row_lists_workspace_bytes(n)
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    hip.check(l.xnrs_build_row_lists(hip.ptr(md), hip.ptr(idd), n, L, hip.ptr(buf[0]), hip.ptr(buf[2]) if with_ids else None,
                                     hip.ptr(buf[1]), hip.ptr(buf[3]) if with_ids else None, hip.ptr(counts), hip.ptr(ws), nws,
                                     hip.stream_ptr(torch.device(DEV))), "xnrs_build_row_lists")
    lm = (m[ids.long()] if with_ids else m).ne(0)
    rows = torch.nonzero(lm.reshape(-1)).squeeze(1)
    news = torch.nonzero(lm.any(dim=1)).squeeze(1)
    kv = (news[:, None] * L + torch.arange(L)[None, :]).reshape(-1)
    assert counts.tolist() == [rows.numel(), kv.numel()]
    assert torch.equal(buf[0, :rows.numel()].cpu().long(), rows)
    assert torch.equal(buf[1, :kv.numel()].cpu().long(), kv)
    assert (buf[0, rows.numel():] == -7).all() and (buf[1, kv.numel():] == -7).all()  # nothing written past the counts
    if with_ids:
        src = ids.long()
        seq = rows // L
        assert torch.equal(buf[2, :rows.numel()].cpu().long(), src[seq] * L + (rows - seq * L))
        assert torch.equal(buf[3, :kv.numel()].cpu().long(), (src[news][:, None] * L + torch.arange(L)[None, :]).reshape(-1))


def _nrms(c, att_dropout):
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    model.train()
    for mod in model.modules():
        if isinstance(mod, layers.MultiHeadAttention):
            mod.dropout.p = att_dropout
    return model


def _step(model, batch, labels, seed=1234):
    torch.manual_seed(seed)  # the attention-dropout seeds come from torch's CPU generator (ops._att_dropout)
    model.zero_grad(set_to_none=True)
    preds = torch.relu(model(batch))
    ue = model.get_user_embeddings(batch)
    loss = torch.nn.functional.mse_loss(preds, batch["targets"]) + 0.1 * contrastive_loss(ue, labels, 0.08)
    loss.backward()
    return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


def _close(g1, g0, tol):
    assert g0.keys() == g1.keys()
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        scale = max(g0[k].abs().max().item(), 1e-3 * gmax)
        e = (g1[k] - g0[k]).abs().max().item() / scale
        assert e <= tol, f"{k}: {e:.3e}"


@pytest.mark.parametrize("with_dropout", [False, True])
def test_device_row_lists_step_equals_host_list_step(with_dropout, monkeypatch):
    """The NRMS grad step (two history encodes, InfoNCE, backward) with the row lists and their counts on the device against
    the same step with torch bookkeeping + a host read: same forward bit for bit, gradients equal up to the split-K slicing
    (the device path cuts its K slices from the device count).  35 % empty history slots, S = 50 / d_k = 16."""
    from xnrs_amd import autograd as AG
    c = dict(model="NRMS", B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=False, seed=4101, min_len=3)
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    model = _nrms(c, 0.1 if with_dropout else 0.0)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
    monkeypatch.setattr(AG, "DEVICE_LISTS", False)
    l0, g0 = _step(model, batch, labels)
    monkeypatch.setattr(AG, "DEVICE_LISTS", True)
    before = AG.STATS["device_list_forwards"]
    l1, g1 = _step(model, batch, labels)
    assert AG.STATS["device_list_forwards"] - before >= 2  # history + candidates (the second history encode shares the first's)
    assert torch.equal(l0, l1)
    _close(g1, g0, 2e-5)


@pytest.mark.parametrize("lists", ["device", "host", "dense"])
def test_two_history_encodes_share_one_weight_gradient_product(lists, monkeypatch):
    """dW = (dQKV_1 + dQKV_2)^T . X: the backward nodes of the reference's two history encodes (training.py:406,409; the
    second reads the first one's Q|K|V image) coordinate -- the first to run defers, the second merges -- with attention
    dropout 0.1 ON, so the two encodes really differ.  Gradients equal the unmerged step's up to summation order; when only
    ONE of the two encodes takes part in a backward pass nothing is deferred and its gradients are complete."""
    from xnrs_amd import autograd as AG
    c = dict(model="NRMS", B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=False, seed=4202, min_len=3)
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    monkeypatch.setattr(AG, "LIVE_ROWS", lists != "dense")
    monkeypatch.setattr(AG, "DEVICE_LISTS", lists == "device")
    model = _nrms(c, 0.1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
    monkeypatch.setattr(AG, "MERGE_DW", False)
    l0, g0 = _step(model, batch, labels)
    monkeypatch.setattr(AG, "MERGE_DW", True)
    d0, m0 = AG.STATS["deferred_dqkv_backwards"], AG.STATS["merged_dqkv_backwards"]
    l1, g1 = _step(model, batch, labels)
    assert (AG.STATS["deferred_dqkv_backwards"] - d0, AG.STATS["merged_dqkv_backwards"] - m0) == (1, 1)
    assert torch.equal(l0, l1)
    _close(g1, g0, 2e-5)
    # only the InfoNCE branch in the backward pass: its node's partner will not run -> no deferral, complete gradients
    torch.manual_seed(99)
    model.zero_grad(set_to_none=True)
    preds = model(batch)
    ue = model.get_user_embeddings(batch)
    d0 = AG.STATS["deferred_dqkv_backwards"]
    contrastive_loss(ue, labels, 0.08).backward()
    assert AG.STATS["deferred_dqkv_backwards"] == d0
    ga = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    monkeypatch.setattr(AG, "MERGE_DW", False)
    torch.manual_seed(99)
    model.zero_grad(set_to_none=True)
    preds = model(batch)  # noqa: F841
    ue = model.get_user_embeddings(batch)
    contrastive_loss(ue, labels, 0.08).backward()
    gb = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert ga.keys() == gb.keys()
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_merged_step_matches_reference_golden_shipped_shape(monkeypatch):
    """The real reference's train step at the shipped shape (tests/golden/grads_shipped.npz) through device-built row lists
    and the merged weight-gradient product."""
    from xnrs_amd import autograd as AG
    g = H.golden("grads_shipped")
    c = cases.GRAD_SHIPPED
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    # (the golden step was recorded in eval mode: no dropout, so the second encode would be shared WHOLE -- autograd._OUTPUTS;
    # switched off here so that both encodes run and their backwards merge)
    monkeypatch.setattr(AG, "SHARE_OUTPUTS", False)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = cases.theme_labels(c["themes"]).to(DEV)
    d0, m0, v0 = AG.STATS["deferred_dqkv_backwards"], AG.STATS["merged_dqkv_backwards"], AG.STATS["device_list_forwards"]
    preds = torch.relu(model(batch))
    loss_cl = contrastive_loss(model.get_user_embeddings(batch), labels, c["temperature"])
    loss = torch.nn.functional.mse_loss(preds, batch["targets"]) + c["lambda_cl"] * loss_cl
    loss.backward()
    assert AG.STATS["deferred_dqkv_backwards"] == d0 + 1 and AG.STATS["merged_dqkv_backwards"] == m0 + 1
    assert AG.STATS["device_list_forwards"] >= v0 + 2
    H.assert_close(loss, g["gs/loss"], 1e-5)
    n = H.assert_sampled_grads_close({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, 2e-4)
    assert n >= 28


def test_grad_step_is_captured_in_a_hipgraph_and_replays_bitwise(monkeypatch):
    """forward + relu/MSE + InfoNCE on the second history encode + backward of NRMS (attention dropout 0.1, 30 % empty history
    slots) has no host synchronisation: torch.cuda.graph captures it, and replays give bit for bit the eager step's loss
    and gradients -- also after the inputs were overwritten in place with another batch (the device-built row lists follow
    the new mask)."""
    from xnrs_amd import autograd as AG
    c = dict(model="NRMS", B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=False, seed=4303, min_len=3)
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    model = _nrms(c, 0.1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    c2 = dict(c, seed=4404)
    batch2 = synth.batch_to(cases.model_batch(c2), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        p.grad = torch.zeros_like(p)

    def step():
        for p in params:
            p.grad.zero_()
        preds = torch.relu(model(batch))
        ue = model.get_user_embeddings(batch)
        loss = torch.nn.functional.mse_loss(preds, batch["targets"]) + 0.1 * contrastive_loss(ue, labels, 0.08)
        loss.backward()
        return loss

    def eager():
        torch.manual_seed(77)
        loss = step()
        return loss.detach().clone(), [p.grad.clone() for p in params]

    # Eager steps AND the capture run on ONE side stream (torch.cuda.graph(..., stream=side)).  The engine may reuse a
    # parameter's AccumulateGrad node from an earlier step, with the stream it was created on; when that is not the capture
    # stream the gradient accumulation is captured as a forked branch and the allocator's reuse of freed blocks (which assumes
    # stream order) corrupts the replay (torch warns "AccumulateGrad node's stream does not match"; measured here: every
    # replay wrong with torch's own capture stream, 64 of 64 right on the warm-up stream).
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        l0, g0 = eager()
        torch.manual_seed(77)
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    torch.manual_seed(77)
    with torch.cuda.graph(graph, stream=side):
        loss_g = step()
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(loss_g, l0)
        for p, ref in zip(params, g0):
            assert torch.equal(p.grad, ref)
    # another batch written into the captured input tensors: the replay must equal the eager step on that batch
    def copy_batch(dst, src):
        for feat in ("user_features", "candidate_features"):
            d = dst[feat]["history"]["title_emb"] if feat == "user_features" else dst[feat]["title_emb"]
            s = src[feat]["history"]["title_emb"] if feat == "user_features" else src[feat]["title_emb"]
            d[0].copy_(s[0])
            d[1].copy_(s[1])
    copy_batch(batch, batch2)
    graph.replay()
    torch.cuda.synchronize()
    lg = loss_g.clone()
    gg = [p.grad.clone() for p in params]
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        l1, g1 = eager()
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(lg, l1) and not torch.equal(l1, l0)
    for a, b in zip(gg, g1):
        assert torch.equal(a, b)


def test_standardrec_second_encode_is_the_first_bit_for_bit(monkeypatch):
    """StandardRec has no dropout anywhere (mind_small_CL.yml: news / user dropout 0, no attention), so in TRAIN mode the
    reference's second history encode, get_user_embeddings(batch) (parent.py:49-81, training.py:409), is bit for bit the u of
    forward(batch, return_embeddings=True).  The library therefore does not compute it: a deterministic encode called again
    with the very same tensors returns the first call's output tensors (autograd._OUTPUTS) -- news tower AND user tower --
    and the step's gradients equal those of the step that encodes twice."""
    from xnrs_amd import autograd as AG
    c = dict(cases.GRAD_SHIPPED_STD, B=6, H=7, C=3, seed=4505)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    model.train()
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0], device=DEV)

    def step(share):
        monkeypatch.setattr(AG, "SHARE_OUTPUTS", share)
        model.zero_grad(set_to_none=True)
        before = AG.STATS["shared_output_forwards"]
        r, u, _ = model(batch, return_embeddings=True)
        ue = model.get_user_embeddings(batch)
        took = AG.STATS["shared_output_forwards"] - before
        assert torch.equal(ue, u.squeeze(1))
        loss = torch.nn.functional.mse_loss(torch.relu(r), batch["targets"]) + 0.1 * contrastive_loss(ue, labels, 0.08)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, took

    l0, g0, t0 = step(False)
    l1, g1, t1 = step(True)
    assert (t0, t1) == (0, 2)  # the second history encode and the second user encode
    assert torch.equal(l0, l1)
    _close(g1, g0, 2e-5)
    # a new step is a new graph: nothing is reused across steps (the first call's graph is gone after its backward) ...
    before = AG.STATS["shared_output_forwards"]
    r = model(batch)
    assert AG.STATS["shared_output_forwards"] == before
    # ... and an in-place change of the input between the two calls prevents the reuse
    batch["user_features"]["history"]["title_emb"][0].mul_(1.0)
    model.get_user_embeddings(batch)
    assert AG.STATS["shared_output_forwards"] == before
    del r


def test_shared_projection_is_bitwise_neutral_with_attention_dropout_on(monkeypatch):
    """The shipped train mode (attention dropout 0.1): the second history encode reading the first one's Q|K|V image
    (autograd._QKV_IMAGES) gives bit for bit the loss and gradients of the step that projects twice -- the two encodes draw
    different dropout masks, their projections are the same numbers.  (Merged weight gradients off: they change the
    summation order and have their own test above.)"""
    from xnrs_amd import autograd as AG
    c = dict(model="NRMS", B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=False, seed=4606, min_len=3)
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    monkeypatch.setattr(AG, "MERGE_DW", False)
    model = _nrms(c, 0.1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
    monkeypatch.setattr(AG, "SHARE_QKV", False)
    l0, g0 = _step(model, batch, labels)
    monkeypatch.setattr(AG, "SHARE_QKV", True)
    before = AG.STATS["shared_qkv_forwards"]
    l1, g1 = _step(model, batch, labels)
    assert AG.STATS["shared_qkv_forwards"] == before + 1
    assert torch.equal(l0, l1) and g0.keys() == g1.keys()
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k



@pytest.mark.parametrize("model_name", ["NRMS", "standard"])
def test_side_lane_of_the_backward_is_bitwise_neutral(model_name, monkeypatch):
    """The backward's weight-gradient launches on the library's side stream (api.hip SideLane: forked from and joined back
    into the caller's stream inside the call) against the same step on one stream: the same launches, so loss and every
    gradient bit for bit -- also when the caller's stream is not the default one, and with nothing but the call's own join
    between the backward and the gradients' first reader (the clone right behind it)."""
    c = dict(model=model_name, B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=False, seed=4707, min_len=3)
    from xnrs_amd import autograd as AG
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    model = _nrms(c, 0.1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
    with hip.knobs(XNRS_BWD_SIDE_STREAM="0"):
        l0, g0 = _step(model, batch, labels)
    with hip.knobs(XNRS_BWD_SIDE_STREAM="1", XNRS_BWD_SIDE_MIN_ROWS="0"):
        for rep in range(3):  # (repeated: a missing join would show as a stale or half-written gradient now and then)
            l1, g1 = _step(model, batch, labels)
            assert torch.equal(l0, l1) and g0.keys() == g1.keys()
            for k in g0:
                assert torch.equal(g0[k], g1[k]), (rep, k)
        side = torch.cuda.Stream(device=DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            l2, g2 = _step(model, batch, labels)
        side.synchronize()
        assert torch.equal(l0, l2)
        for k in g0:
            assert torch.equal(g0[k], g2[k]), k


def test_folded_pair_is_computed_once_per_tower_and_weight_version(monkeypatch):
    """autograd._train_fold: the W1.Wo / W1.bo + b1 pair of a training forward comes from a per-(tower, weight version) cache
    and is handed to the forward and its backward (xnrs_additive_params.w1_folded): two optimizer steps with the cache equal
    the same two steps with every call folding for itself, bit for bit, and the cache really was hit."""
    from xnrs_amd import autograd as AG
    c = dict(model="NRMS", B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=True, seed=4808, min_len=3)
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)

    def two_steps(cache):
        monkeypatch.setattr(AG, "FOLD_TRAIN_CACHE", cache)
        AG._TRAIN_FOLDS.clear()
        model = _nrms(c, 0.1)
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        out = []
        for it in range(2):
            loss, grads = _step(model, batch, labels, seed=77 + it)
            opt.step()
            out.append((loss, grads))
        return out, {k: v.detach().clone() for k, v in model.state_dict().items()}
    real = AG._train_fold
    calls = {"n": 0, "miss": 0}

    def spy(*a):
        before = len(AG._TRAIN_FOLDS)
        r = real(*a)
        if r is not None:
            calls["n"] += 1
            calls["miss"] += int(len(AG._TRAIN_FOLDS) != before)
        return r
    monkeypatch.setattr(AG, "_train_fold", spy)
    with_cache, sd1 = two_steps(True)
    assert calls["n"] == 10 and calls["miss"] == 4, calls   # 5 tower calls per step, 2 towers: 2 folds per step instead of 5
    without, sd0 = two_steps(False)
    for (l1, g1), (l0, g0) in zip(with_cache, without):
        assert torch.equal(l1, l0) and g1.keys() == g0.keys()
        for k in g0:
            assert torch.equal(g1[k], g0[k]), k
    for k in sd0:
        assert torch.equal(sd1[k], sd0[k]), k


@pytest.mark.parametrize("model_name", ["NRMS", "standard"])
def test_shared_parameter_gradients_are_summed_by_the_last_node(model_name, monkeypatch):
    """autograd._sum_with_group: the calls of a tower that are not its last in a backward pass keep their parameter gradients
    back and the last one adds them with one multi-tensor launch per kept set -- the gradients autograd's pairwise adds give,
    up to the order of the additions (fp32 rounding), and the input gradient / loss bit for bit."""
    from xnrs_amd import autograd as AG
    c = dict(model=model_name, B=8, H=12, C=3, S=50, D=64, h=4, E=32, bias=True, seed=4909, min_len=3)
    monkeypatch.setattr(AG, "LIVE_ROWS_MIN", 1)
    model = _nrms(c, 0.1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0, 2, 2], device=DEV)
    monkeypatch.setattr(AG, "SUM_SHARED_GRADS", False)
    l0, g0 = _step(model, batch, labels)
    monkeypatch.setattr(AG, "SUM_SHARED_GRADS", True)
    kept = []
    real = AG._sum_with_group

    def spy(ctx, grads):
        out = real(ctx, grads)
        kept.append(all(t is None for t in out) and any(t is not None for t in grads))
        return out
    monkeypatch.setattr(AG, "_sum_with_group", spy)
    l1, g1 = _step(model, batch, labels)
    assert any(kept) and not all(kept)       # some node kept its set back, some node delivered
    assert torch.equal(l0, l1)
    _close(g1, g0, 2e-6)
    assert all(not g.kept for g in AG._PARAM_GROUPS.values())  # nothing is left behind after a pass
