"""GPU: gradients of the HIP path (hand-written backward kernels through torch.autograd.Function)
against (a) the golden gradients recorded from the real reference's train step and (b) torch
autograd through the CPU oracle on the same seeded inputs."""
import pytest
import torch

from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases
from xnrs_amd import synth
from xnrs_amd.losses import contrastive_loss
from xnrs_amd.models import make_model
from xnrs_amd.models.components import layers, news_encoding, user_encoding

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GTOL = 2e-4  # gradients: relative to max(|ref| of the tensor, 1e-3 * largest gradient of the model)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def load(module, seed, train=False):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed)
    module.load_state_dict(sd)
    module.train(train)
    return module.to(DEV), sd


def oracle_sd(sd):
    return {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in sd.items()}


def check_param_grads(module, osd, tol=GTOL):
    gmax = max(v.grad.abs().max().item() for v in osd.values() if v.grad is not None)
    n = 0
    for k, p in module.named_parameters():
        if k.endswith("dummy_param"):
            continue
        ref = osd[k].grad
        assert p.grad is not None, f"no grad for {k}"
        assert ref is not None, k
        scale = max(ref.abs().max().item(), 1e-3 * gmax)
        e = (p.grad.cpu().double() - ref.double()).abs().max().item() / scale
        assert e <= tol, f"{k}: {e:.3e}"
        n += 1
    return n


def test_train_step_matches_reference_golden():
    """Loss + all parameter and input gradients of the reference's train step (training.py:402-472,
    eval-mode dropout) for the tiny NRMS -- golden vectors come from the real reference."""
    g = H.golden("grads")
    c = cases.GRAD
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx = hx.to(DEV).requires_grad_(True)
    cx = cx.to(DEV).requires_grad_(True)
    batch["user_features"]["history"]["title_emb"] = (hx, hm)
    batch["candidate_features"]["title_emb"] = (cx, cm)
    labels = cases.theme_labels(batch["main_theme"]).to(DEV)
    preds = torch.relu(model(batch))
    loss_rec = torch.nn.functional.mse_loss(preds, batch["targets"].to(DEV))
    ue = model.get_user_embeddings(batch)
    loss_cl = O.contrastive_loss(ue, labels, c["temperature"])
    loss = loss_rec + c["lambda_cl"] * loss_cl
    loss.backward()
    H.assert_close(loss, g["grad/loss"], 1e-5)
    H.assert_close(hx.grad, g["grad/d_hist_x"], GTOL, "d_hist_x")
    H.assert_close(cx.grad, g["grad/d_cand_x"], GTOL, "d_cand_x")
    n = H.assert_grads_close({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, GTOL)
    assert n >= 28


@pytest.mark.parametrize("live", [True, False])
def test_train_step_matches_reference_golden_shipped_shape(live, monkeypatch):
    """The same train step at the SHIPPED shape (S=50, D=768, 16 heads: d_k = 48, E=256; cases.GRAD_SHIPPED): loss, input
    gradients and all 28 parameter gradients of the REAL reference (whole, or a fixed 4 096-element sample of the big
    tensors) -- mha_bwd_fused_kernel<3>, the folded out-projection backward and, with `live` (the threshold lowered so
    this 250-row batch takes it), the live-row forward / dW path."""
    from xnrs_amd import autograd
    g = H.golden("grads_shipped")
    c = cases.GRAD_SHIPPED
    monkeypatch.setattr(autograd, "LIVE_ROWS", live)
    monkeypatch.setattr(autograd, "LIVE_ROWS_MIN", 1)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx = hx.to(DEV).requires_grad_(True)
    cx = cx.to(DEV).requires_grad_(True)
    batch["user_features"]["history"]["title_emb"] = (hx, hm)
    batch["candidate_features"]["title_emb"] = (cx, cm)
    labels = cases.theme_labels(c["themes"]).to(DEV)
    preds = torch.relu(model(batch))
    loss_rec = torch.nn.functional.mse_loss(preds, batch["targets"].to(DEV))
    loss_cl = contrastive_loss(model.get_user_embeddings(batch), labels, c["temperature"])  # the fused HIP InfoNCE
    loss = loss_rec + c["lambda_cl"] * loss_cl
    loss.backward()
    H.assert_close(loss, g["gs/loss"], 1e-5)
    H.assert_close(loss_cl, g["gs/loss_cl"], 1e-5)
    H.assert_close(cases.grad_sample(hx.grad), g["gs/d_hist_x"], GTOL, "d_hist_x")
    H.assert_close(cases.grad_sample(cx.grad), g["gs/d_cand_x"], GTOL, "d_cand_x")
    n = H.assert_sampled_grads_close({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, GTOL)
    assert n >= 28


@pytest.mark.parametrize("live", [True, False])
def test_standardrec_train_step_matches_reference_golden_shipped_shape(live, monkeypatch):
    """BASELINE configs[3]'s model (StandardRec: attention-free additive towers + heads, biases on) through the same train
    step at the shipped token shape (cases.GRAD_SHIPPED_STD): loss, input gradients and every parameter gradient of the
    REAL reference -- with `live`, fc1 forward / dW1 / the input gradient's fc1 term run over the unmasked token rows."""
    from xnrs_amd import autograd
    g = H.golden("grads_shipped_standard")
    c = cases.GRAD_SHIPPED_STD
    monkeypatch.setattr(autograd, "LIVE_ROWS", live)
    monkeypatch.setattr(autograd, "LIVE_ROWS_MIN", 1)
    before = autograd.STATS["live_row_forwards"]
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx = hx.to(DEV).requires_grad_(True)
    cx = cx.to(DEV).requires_grad_(True)
    batch["user_features"]["history"]["title_emb"] = (hx, hm)
    batch["candidate_features"]["title_emb"] = (cx, cm)
    labels = cases.theme_labels(c["themes"]).to(DEV)
    preds = torch.relu(model(batch))
    loss_rec = torch.nn.functional.mse_loss(preds, batch["targets"].to(DEV))
    loss_cl = contrastive_loss(model.get_user_embeddings(batch), labels, c["temperature"])
    loss = loss_rec + c["lambda_cl"] * loss_cl
    loss.backward()
    assert (autograd.STATS["live_row_forwards"] > before) == live
    H.assert_close(loss, g["gss/loss"], 1e-5)
    H.assert_close(loss_cl, g["gss/loss_cl"], 1e-5)
    H.assert_close(cases.grad_sample(hx.grad), g["gss/d_hist_x"], GTOL, "d_hist_x")
    H.assert_close(cases.grad_sample(cx.grad), g["gss/d_cand_x"], GTOL, "d_cand_x")
    n = H.assert_sampled_grads_close({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, GTOL,
                                     "gss/dW/", "gss/max/")
    assert n >= 16


@pytest.mark.parametrize("live", [True, False])
def test_naml_train_step_matches_reference_golden_shipped_shape(live, monkeypatch):
    """BASELINE configs[4]'s model (NAML) through the train step at the shipped token shape (cases.GRAD_SHIPPED_NAML): loss,
    title-token input gradients and all 30 parameter gradients of the REAL reference, incl. the category / subcategory
    embedding tables (deterministic scatter) and the feature pooler."""
    from xnrs_amd import autograd
    g = H.golden("grads_shipped_naml")
    c = cases.GRAD_SHIPPED_NAML
    monkeypatch.setattr(autograd, "LIVE_ROWS", live)
    monkeypatch.setattr(autograd, "LIVE_ROWS_MIN", 1)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx = hx.requires_grad_(True)
    cx = cx.requires_grad_(True)
    batch["user_features"]["history"]["title_emb"] = (hx, hm)
    batch["candidate_features"]["title_emb"] = (cx, cm)
    labels = cases.theme_labels(c["themes"]).to(DEV)
    preds = torch.relu(model(batch))
    loss_rec = torch.nn.functional.mse_loss(preds, batch["targets"].to(DEV))
    ue = model.get_user_embeddings(batch)  # (B, 1, E) un-squeezed (naml.py:146-147); the trainer flattens (training.py:443-444)
    loss_cl = contrastive_loss(ue.reshape(ue.shape[0], -1), labels, c["temperature"])
    loss = loss_rec + c["lambda_cl"] * loss_cl
    loss.backward()
    H.assert_close(loss, g["gsn/loss"], 1e-5)
    H.assert_close(loss_cl, g["gsn/loss_cl"], 1e-5)
    H.assert_close(cases.grad_sample(hx.grad), g["gsn/d_hist_x"], GTOL, "d_hist_x")
    H.assert_close(cases.grad_sample(cx.grad), g["gsn/d_cand_x"], GTOL, "d_cand_x")
    n = H.assert_sampled_grads_close({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, GTOL,
                                     "gsn/dW/", "gsn/max/")
    assert n >= 30


@pytest.mark.parametrize("S,D,h", [(8, 32, 4), (30, 300, 15), (50, 64, 4), (9, 18, 3)])
def test_mha_grads(S, D, h):
    att, sd = load(layers.MultiHeadAttention(h, D), 41)
    rng = synth.rng_for(42)
    x = torch.from_numpy(rng.standard_normal((3, S, D)).astype("float32"))
    m = torch.from_numpy(cases.block_mask(rng, 3, S))
    w = torch.from_numpy(rng.standard_normal((3, S, D)).astype("float32"))
    xd = x.to(DEV).requires_grad_(True)
    y = att(xd, m.to(DEV))
    (y * w.to(DEV)).sum().backward()
    osd = oracle_sd(sd)
    xo = x.clone().requires_grad_(True)
    yo = O.multi_head_attention(xo, m, osd, h)
    (yo * w).sum().backward()
    H.assert_close(y, yo, what="fwd")
    H.assert_close(xd.grad, xo.grad, GTOL, "dx")
    assert check_param_grads(att, osd) == 8


@pytest.mark.parametrize("mask", [True, False])
def test_additive_grads(mask):
    pool, sd = load(layers.AdditiveAttention(48, 256), 43)
    rng = synth.rng_for(44)
    x = torch.from_numpy(rng.standard_normal((4, 11, 48)).astype("float32"))
    m = torch.from_numpy(cases.block_mask(rng, 4, 11)) if mask else None
    w = torch.from_numpy(rng.standard_normal((4, 1, 48)).astype("float32"))
    xd = x.to(DEV).requires_grad_(True)
    y = pool(xd, None if m is None else m.to(DEV))
    (y * w.to(DEV)).sum().backward()
    osd = oracle_sd(sd)
    xo = x.clone().requires_grad_(True)
    yo = O.additive_attention(xo, m, osd)
    (yo * w).sum().backward()
    H.assert_close(y, yo)
    H.assert_close(xd.grad, xo.grad, GTOL, "dx")
    assert check_param_grads(pool, osd) == 4


def test_masked_mean_grad():
    rng = synth.rng_for(45)
    x = torch.from_numpy(rng.standard_normal((3, 7, 20)).astype("float32"))
    m = torch.from_numpy(cases.block_mask(rng, 3, 7))
    xd = x.to(DEV).requires_grad_(True)
    y = layers.MaskedMean()(xd, m.to(DEV))
    y.pow(2).sum().backward()
    xo = x.clone().requires_grad_(True)
    O.masked_mean(xo, m).pow(2).sum().backward()
    H.assert_close(xd.grad, xo.grad, GTOL)


@pytest.mark.parametrize("act", ["tanh", "identity"])
def test_head_activation_other_than_relu(act):
    """TextEncoder(activation=...) (news_encoding.py:10-18,27-31): nn.Tanh and nn.Identity heads, forward and gradients
    against torch autograd of the same stack on the CPU; any other activation module is refused at construction."""
    import torch.nn as nn
    mk = {"tanh": nn.Tanh, "identity": nn.Identity}[act]
    S, D, E = 7, 20, 12
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 24), p_dropout=0.0, out_features=E, in_features=D,
                                             head=True, activation=mk(), att=None, bias=True), 47)
    rng = synth.rng_for(48)
    x, m = synth.token_block(rng, 2, 3, S, D, min_len=1)
    xd = x.to(DEV).requires_grad_(True)
    y, _ = enc((xd, m.to(DEV)))
    y.pow(2).sum().backward()
    osd = oracle_sd(sd)
    xo = x.clone().requires_grad_(True)
    pooled = O.additive_attention(xo.reshape(6, S, D), m.reshape(6, S, 1), {k[7:]: v for k, v in osd.items() if k.startswith("pooler.")})
    hmid = mk()(torch.nn.functional.linear(pooled.reshape(6, D), osd["head.0.weight"], osd["head.0.bias"]))
    yo = torch.nn.functional.linear(hmid, osd["head.2.weight"], osd["head.2.bias"]).reshape(2, 3, E)
    yo.pow(2).sum().backward()
    H.assert_close(y, yo, what="forward")
    H.assert_close(xd.grad, xo.grad, GTOL, "dx")
    assert check_param_grads(enc, osd) == 8
    with pytest.raises(NotImplementedError):
        news_encoding.TextEncoder(pooler=layers.MaskedMean(), p_dropout=0.0, out_features=E, in_features=D, head=True,
                                  activation=nn.GELU())


@pytest.mark.parametrize("normalize", [False, True])
def test_dot_scoring_grads(normalize):
    """DotScoring (scoring.py:12-23), both settings of `normalize`, against torch autograd of the reference formula:
    value, d/du and d/dc; E not a multiple of the wave width, more candidates than waves."""
    from xnrs_amd.models.components import scoring
    rng = synth.rng_for(46)
    u = torch.from_numpy(rng.standard_normal((3, 1, 50)).astype("float32"))
    c = torch.from_numpy(rng.standard_normal((3, 7, 50)).astype("float32"))
    w = torch.from_numpy(rng.standard_normal((3, 7, 1)).astype("float32"))
    ud, cd = u.to(DEV).requires_grad_(True), c.to(DEV).requires_grad_(True)
    r = scoring.DotScoring(normalize=normalize)(ud, cd)
    (r * w.to(DEV)).sum().backward()
    uo, co = u.clone().requires_grad_(True), c.clone().requires_grad_(True)
    un, cn = (uo / uo.norm(p=2, dim=2, keepdim=True), co / co.norm(p=2, dim=2, keepdim=True)) if normalize else (uo, co)
    ro = torch.bmm(cn, un.transpose(-1, -2))
    (ro * w).sum().backward()
    H.assert_close(r, ro, what="scores")
    H.assert_close(ud.grad, uo.grad, GTOL, "du")
    H.assert_close(cd.grad, co.grad, GTOL, "dc")


@pytest.mark.parametrize("name", ["standard_bias", "standard_tiny", "base_tiny", "nrms_300", "naml_tiny"])
def test_model_grads_vs_oracle(name):
    c = cases.MODELS[name]
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = cases.model_batch(c)
    osd = oracle_sd(sd)
    targets = batch["targets"]
    if c["model"] == "NAML":
        r = model(synth.batch_to(batch, DEV))
        ro = O.naml_forward(batch, osd)
    else:
        r = model(batch)
        ro = O.parent_forward(batch["user_features"]["history"]["title_emb"], batch["candidate_features"]["title_emb"],
                              osd, c["h"])
    torch.nn.functional.mse_loss(torch.relu(r), targets.to(DEV)).backward()
    torch.nn.functional.mse_loss(torch.relu(ro), targets).backward()
    H.assert_close(r, ro)
    n = check_param_grads(model, osd)
    assert n >= 10


def test_attention_dropout_train_mode():
    """Train mode enables the Dropout(0.1) on the attention probabilities (layers.py:117,148).  RNG
    streams cannot match torch's; check determinism per seed, the expectation, and that backward uses
    the same mask (finite-difference on one weight)."""
    att, sd = load(layers.MultiHeadAttention(4, 32), 51, train=True)
    x = torch.randn(64, 16, 32, device=DEV)
    m = torch.ones(64, 16, 1, device=DEV)
    with torch.no_grad():
        torch.manual_seed(1)
        y1 = att(x, m)
        torch.manual_seed(1)
        y2 = att(x, m)
        torch.manual_seed(2)
        y3 = att(x, m)
        att.eval()
        ye = att(x, m)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    ys = torch.zeros_like(ye)
    att.train()
    with torch.no_grad():
        for s in range(200):
            torch.manual_seed(100 + s)
            ys += att(x, m)
    rel = (ys / 200 - ye).abs().mean() / ye.abs().mean()
    assert rel < 0.05, rel
    # backward consistency under dropout: directional finite difference in fp32
    torch.manual_seed(7)
    xg = x[:4].clone().requires_grad_(True)
    y = att(xg, m[:4])
    w = torch.randn_like(y)
    (y * w).sum().backward()
    d = torch.randn_like(xg)
    eps = 1e-2
    with torch.no_grad():
        torch.manual_seed(7)
        yp = att(xg + eps * d, m[:4])
        torch.manual_seed(7)
        ym = att(xg - eps * d, m[:4])
    fd = ((yp - ym) * w).sum() / (2 * eps)
    an = (xg.grad * d).sum()
    assert abs(fd - an) / abs(an) < 2e-2, (fd.item(), an.item())


def test_user_encoder_return_weights_grad():
    enc, sd = load(user_encoding.UserEncoder(pooler=layers.AdditiveAttention(32, 256), p_dropout=0.0, emb_dim=32,
                                             att=layers.MultiHeadAttention(4, 32), head=True, bias=True), 61)
    rng = synth.rng_for(62)
    x = torch.from_numpy(rng.standard_normal((3, 7, 32)).astype("float32"))
    m = torch.from_numpy(cases.block_mask(rng, 3, 7))
    xd = x.to(DEV).requires_grad_(True)
    y, a = enc((xd, m.to(DEV)), None, return_weights=True)
    y.sum().backward()
    osd = oracle_sd(sd)
    xo = x.clone().requires_grad_(True)
    yo, ao = O.user_encoder(xo, m, osd, 4, return_weights=True)
    yo.sum().backward()
    H.assert_close(a, ao)
    H.assert_close(xd.grad, xo.grad, GTOL)
    check_param_grads(enc, osd)


def test_fused_infonce_matches_reference_golden_and_oracle():
    """xnrs_amd.losses.contrastive_loss == the REAL reference's _compute_contrastive_loss (golden) and the
    oracle, value and gradient; rows without a positive are skipped; no-positive batches give 0."""
    from xnrs_amd.losses import contrastive_loss
    g = H.golden("grads")
    e, lab = cases.infonce_inputs()
    ed = e.to(DEV).requires_grad_(True)
    l = contrastive_loss(ed, lab.to(DEV), 0.08)
    l.backward()
    H.assert_close(l, g["infonce/loss"], 2e-6)
    H.assert_close(ed.grad, g["infonce/grad"], 2e-5)
    rng = synth.rng_for(71)
    for B, E, nlab in ((64, 256, 6), (33, 48, 40), (16, 1024, 2), (5, 16, 5)):
        x = torch.from_numpy(rng.standard_normal((B, E)).astype("float32") * 0.3)
        lab = torch.from_numpy(rng.integers(0, nlab, size=(B,)))
        if B == 5:
            lab = torch.arange(5)  # all unique: count == 0 -> loss 0 / 1e-8 = 0
        xd = x.to(DEV).requires_grad_(True)
        ld = contrastive_loss(xd, lab.to(DEV), 0.08)
        (ld * 1.7).backward()
        xo = x.clone().requires_grad_(True)
        lo = O.contrastive_loss(xo, lab, 0.08)
        (lo * 1.7).backward()
        H.assert_close(ld, lo, 2e-5, f"loss B={B}") if lo.abs() > 0 else None
        if lo.abs() == 0:
            assert ld.item() == 0.0 and xd.grad.abs().max().item() == 0.0
        else:
            H.assert_close(xd.grad, xo.grad, 1e-4, f"grad B={B}")
    # NAML hands (B,1,E) (naml.py:146-147)
    x3 = torch.randn(8, 1, 16, device=DEV)
    lab = torch.tensor([0, 1, 0, 1, 2, 2, 0, 1], device=DEV)
    assert torch.equal(contrastive_loss(x3, lab, 0.08), contrastive_loss(x3[:, 0], lab, 0.08))


def test_skip_empty_gradients_match_dense():
    """TextEncoder.skip_empty under autograd: the empty history slots share one encoded representative (it collects
    their gradient, e.g. towards the head biases); loss and parameter gradients equal the dense step."""
    import torch.nn.functional as F
    from tests.golden import cases as cs
    from xnrs_amd.models import make_model

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    c = dict(model="NRMS", E=32, bias=True, h=4, D=32, H=10, S=6)
    torch.manual_seed(3)
    model = make_model(Cfg(cs.model_cfg(c))).to(DEV).eval()  # eval: no dropout, so both runs are deterministic
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    B, Hh, C, S, D = 12, 10, 4, 6, 32
    hx = torch.randn(B, Hh, S, D, device=DEV, generator=g)
    hm = (torch.rand(B, Hh, S, 1, device=DEV, generator=g) < 0.7).float()
    hm[:, 6:] = 0          # trailing empty slots ...
    hx[:, 6:] = 0
    hx[0, 7] = 3.0         # ... one of them with non-zero tokens
    cx = torch.randn(B, C, S, D, device=DEV, generator=g)
    cm = torch.ones(B, C, S, 1, device=DEV)
    tgt = torch.zeros(B, C, 1, device=DEV)
    tgt[:, 0] = 1

    def run(flag):
        model.zero_grad(set_to_none=True)
        model.news_encoder.skip_empty = flag
        try:
            r = model._forward((hx, hm), (cx, cm))
            loss = F.mse_loss(r, tgt)  # no relu: random-init scores may all be negative
            loss.backward()
        finally:
            model.news_encoder.skip_empty = False
        return loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert abs(l0.item() - l1.item()) <= 1e-6 * max(1.0, abs(l0.item()))
    assert g0.keys() == g1.keys() and len(g0) > 10
    gmax = max(v.abs().max().item() for v in g0.values())
    assert gmax > 0
    for k, ref in g0.items():
        scale = max(ref.abs().max().item(), 1e-3 * gmax)
        assert (g1[k] - ref).abs().max().item() / scale <= 1e-4, k


@pytest.mark.parametrize("tower", ["attention", "additive_only"])
@pytest.mark.parametrize("with_ids", [False, True])
def test_backward_over_live_rows_matches_dense_backward(with_ids, tower):
    """xnrs_seq_encoder_bwd_live (the row-parallel backward products over the unmasked token rows only) against the
    dense backward and against oracle autograd: parameter gradients of a TextEncoder with attention, masks with
    holes and fully masked news, with and without the id-gather (table) path, and the input gradient."""
    from xnrs_amd import autograd as AG
    S, D, h, E = 24, 64, 4, 32
    att = layers.MultiHeadAttention(h, D) if tower == "attention" else None  # (StandardRec / NAML views: fc1 over live rows)
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 48), p_dropout=0.0, out_features=E,
                                             in_features=D, att=att), 171)
    rng = synth.rng_for(172)
    n_tab = 260
    x = torch.from_numpy(rng.standard_normal((n_tab, S, D)).astype("float32"))
    m = torch.from_numpy((rng.random((n_tab, S)) < 0.55).astype("float32"))
    m[:5] = 0
    m[torch.from_numpy(rng.random(n_tab) < 0.3)] = 0  # empty history slots: the K|V row list (xnrs_row_lists) leaves them out
    w = torch.from_numpy(rng.standard_normal((n_tab if not with_ids else 300, E)).astype("float32"))
    ids = torch.from_numpy(rng.integers(0, n_tab, size=(300,)).astype("int64")) if with_ids else None

    def run(live, kv=True):
        AG.LIVE_ROWS = live
        AG.KV_ROWS = kv
        try:
            enc.zero_grad(set_to_none=True)
            xd = x.to(DEV).requires_grad_(not with_ids)
            if with_ids:
                y, _ = enc.forward_ids(xd, m.to(DEV), ids.to(DEV).reshape(1, -1))
                y = y[0]
            else:
                y, _ = enc((xd.unsqueeze(0), m.to(DEV).reshape(1, n_tab, S, 1)))
                y = y[0]
            (y * w.to(DEV)).sum().backward()
        finally:
            AG.LIVE_ROWS = True
            AG.KV_ROWS = True
        return y.detach(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}, xd.grad

    y0, g0, dx0 = run(False)
    before = AG.STATS["kv_row_forwards"]
    y1, g1, dx1 = run(True)
    assert AG.STATS["kv_row_forwards"] == before + (1 if att is not None else 0)  # the K|V list was in use
    y2, g2, dx2 = run(True, kv=False)
    assert AG.STATS["kv_row_forwards"] == before + (1 if att is not None else 0)
    assert torch.equal(y0, y1) and torch.equal(y0, y2) and g0.keys() == g1.keys() == g2.keys()
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        scale = max(g0[k].abs().max().item(), 1e-3 * gmax)
        assert (g1[k] - g0[k]).abs().max().item() / scale <= 2e-5, k
        assert (g2[k] - g0[k]).abs().max().item() / scale <= 2e-5, k
    if not with_ids:
        H.assert_close(dx1, dx0, 2e-5, "dx live vs dense")
        H.assert_close(dx2, dx0, 2e-5, "dx live (dense K|V) vs dense")
        # and against torch autograd through the oracle
        osd = oracle_sd(sd)
        xo = x.clone().requires_grad_(True)
        yo, _ = O.text_encoder(xo.unsqueeze(0), m.reshape(1, n_tab, S, 1), osd, h)
        (yo[0] * w).sum().backward()
        H.assert_close(dx1, xo.grad, GTOL, "dx vs oracle")
        assert check_param_grads(enc, osd) >= (12 if att is not None else 6)


@pytest.mark.parametrize("D", [96, 448])
@pytest.mark.parametrize("with_ids", [False, True])
@pytest.mark.parametrize("live", [False, True])
def test_weight_gradient_kernels_agree(with_ids, live, D):
    """The register-transposing weight-gradient kernels (gemm_dw.hip; XNRS_GEMM_DW=3 sends every eligible launch to
    them: dense rows, live-row lists, table-gathered X) against the generic k-major kernel (XNRS_GEMM_DW=0) on a
    contraction long enough to qualify (>= 8192 rows), widths that leave a ragged tile (D = 96: the 128 x 128 tile;
    D = 448 = 256 + 192: the 256 x 256 tile, and with XNRS_GEMM_DW_TILE=128 the small one on the same shapes), and
    a split-K count that is not a multiple of the 8 XCDs."""
    from xnrs_amd import autograd as AG, hip
    S, h, E = 24, 4, 32
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 80), p_dropout=0.0, out_features=E,
                                             in_features=D, att=layers.MultiHeadAttention(h, D)), 181)
    rng = synth.rng_for(182)
    n_tab, n_news = 900, 1000
    x = torch.from_numpy(rng.standard_normal((n_tab, S, D)).astype("float32"))
    m = torch.from_numpy((rng.random((n_tab, S)) < 0.6).astype("float32"))
    m[:7] = 0
    w = torch.from_numpy(rng.standard_normal((n_news if with_ids else n_tab, E)).astype("float32")).to(DEV)
    ids = torch.from_numpy(rng.integers(0, n_tab, size=(n_news,)).astype("int64")) if with_ids else None

    def run(knob, tile="256"):
        with hip.knobs(XNRS_GEMM_DW=knob, XNRS_GEMM_DW_TILE=tile):
            AG.LIVE_ROWS = live
            try:
                enc.zero_grad(set_to_none=True)
                xd = x.to(DEV)
                if with_ids:
                    y, _ = enc.forward_ids(xd, m.to(DEV), ids.to(DEV).reshape(1, -1))
                else:
                    y, _ = enc((xd.unsqueeze(0), m.to(DEV).reshape(1, n_tab, S, 1)))
                (y[0] * w).sum().backward()
            finally:
                AG.LIVE_ROWS = True
        return {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}

    g0 = run("0")
    gmax = max(v.abs().max().item() for v in g0.values())
    for g2 in ([run("3")] if D < 384 else [run("3"), run("3", "128"), run("2")]):
        assert g0.keys() == g2.keys() and len(g0) >= 12
        for k in g0:  # (the key bias gradient is exactly zero in exact arithmetic: softmax shift invariance -> floor the scale)
            scale = max(g0[k].abs().max().item(), 1e-3 * gmax)
            assert (g2[k] - g0[k]).abs().max().item() / scale <= 2e-5, f"{k}: gemm_dw vs k-major kernel"
    if not with_ids:
        osd = oracle_sd(sd)
        yo, _ = O.text_encoder(x.unsqueeze(0), m.reshape(1, n_tab, S, 1), osd, h)
        (yo[0] * w.cpu()).sum().backward()
        assert check_param_grads(enc, osd) >= 12


@pytest.mark.parametrize("S,h,dk", [(33, 2, 48), (34, 3, 20), (36, 2, 64), (49, 2, 48), (50, 4, 48), (51, 2, 16),
                                    (52, 3, 36), (37, 2, 48), (48, 2, 64), (53, 2, 20), (64, 2, 48), (50, 2, 4)])
def test_attention_core_pair_kernel_branches(S, h, dk):
    """mha_core_pair_kernel (mha_core.hip): every (full key tiles, tail keys) split -- S = 16 KTM + 1..4 goes through
    the VALU tail, anything else through padded tiles -- times every feature-block count (d_k = 4 .. 64, with and
    without feature padding), forward AND backward (the forward under grad keeps the softmax row statistics the
    backward recomputes P from), against the oracle; and bitwise against the first-generation kernel where the two share their
    arithmetic (no tail)."""
    from xnrs_amd import hip
    D = h * dk
    att, sd = load(layers.MultiHeadAttention(h, D), 61)  # eval mode: no attention dropout; grads on: statistics kept
    rng = synth.rng_for(62 + S)
    n = 5
    x = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32"))
    m = torch.from_numpy(cases.block_mask(rng, n, S))
    m[0] = 0  # a fully masked sequence: uniform 1/S rows over ALL keys, tail included
    w = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32"))
    xd = x.to(DEV).requires_grad_(True)
    y = att(xd, m.to(DEV))
    (y * w.to(DEV)).sum().backward()
    osd = oracle_sd(sd)
    xo = x.clone().requires_grad_(True)
    yo = O.multi_head_attention(xo, m, osd, h)
    (yo * w).sum().backward()
    H.assert_close(y, yo, what="fwd")
    H.assert_close(xd.grad, xo.grad, GTOL, "dx")
    assert check_param_grads(att, osd) == 8
    with torch.no_grad(), hip.knobs(XNRS_MHA_PAIR="0"):
        y_old = att.eval()(x.to(DEV), m.to(DEV))
    with torch.no_grad():
        y_new = att.eval()(x.to(DEV), m.to(DEV))
    if 1 <= S % 16 <= 4:
        H.assert_close(y_new, y_old, 2e-6, "tail kernel vs first-generation kernel")
    else:
        assert torch.equal(y_new, y_old)


@pytest.mark.parametrize("live,bias", [(True, True), (False, True), (True, False)])
def test_folded_out_projection_gradients(live, bias):
    """Training with the out-projection folded behind the pooling (api.hip "fold": the forward pools the O rows, the
    backward builds dO = a_i g + dpre W', dW1 = dW' Wo^T + db' (x) bo, dWo = dp^T po + W1^T dW', dbo = sum s dp + W1^T db')
    against the per-token order (XNRS_FOLD_TRAIN=0) and oracle autograd: every parameter gradient and the input
    gradient, with and without biases, over live rows and dense, masks with holes, an all-masked and a fully live news."""
    from xnrs_amd import autograd as AG, hip
    S, D, h, E, A = 24, 64, 4, 32, 48
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=E,
                                             in_features=D, att=layers.MultiHeadAttention(h, D), bias=bias), 191)
    rng = synth.rng_for(192)
    n = 150
    x = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32"))
    m = torch.from_numpy((rng.random((n, S)) < 0.6).astype("float32"))
    m[:3] = 0
    m[3] = 1
    w = torch.from_numpy(rng.standard_normal((n, E)).astype("float32")).to(DEV)

    took_live = []

    def run(knob):
        with hip.knobs(XNRS_FOLD_TRAIN=knob):
            AG.LIVE_ROWS = live
            old_min, AG.LIVE_ROWS_MIN = AG.LIVE_ROWS_MIN, 1  # 150 x 24 = 3 600 rows: below the production threshold of 4 096
            before = AG.STATS["live_row_forwards"]
            try:
                enc.zero_grad(set_to_none=True)
                xd = x.to(DEV).requires_grad_(True)
                y, _ = enc((xd.unsqueeze(0), m.to(DEV).reshape(1, n, S, 1)))
                (y[0] * w).sum().backward()
            finally:
                AG.LIVE_ROWS = True
                AG.LIVE_ROWS_MIN = old_min
            took_live.append(AG.STATS["live_row_forwards"] > before)
        return y.detach(), xd.grad, {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}

    y0, dx0, g0 = run("0")
    y1, dx1, g1 = run("2")
    assert all(took_live) == live and any(took_live) == live  # the live-row branch really ran (or really did not)
    assert not torch.equal(y1, y0)  # (really two different computations)
    H.assert_close(y1, y0, 2e-5, "forward, folded vs per-token")
    H.assert_close(dx1, dx0, 5e-5, "dx, folded vs per-token")
    assert g0.keys() == g1.keys() and len(g0) >= (12 if bias else 7)
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        scale = max(g0[k].abs().max().item(), 1e-3 * gmax)
        assert (g1[k] - g0[k]).abs().max().item() / scale <= 5e-5, f"{k}: folded vs per-token gradient"
    osd = oracle_sd(sd)
    xo = x.clone().requires_grad_(True)
    yo, _ = O.text_encoder(xo.unsqueeze(0), m.reshape(1, n, S, 1), osd, h)
    (yo[0] * w.cpu()).sum().backward()
    H.assert_close(dx1, xo.grad, GTOL, "dx vs oracle")
    assert check_param_grads(enc, osd) >= (12 if bias else 7)


def test_second_history_encode_reuses_the_first_projection():
    """The reference's train step encodes the history twice (training.py:406 model(batch), :409 get_user_embeddings(batch)).
    With input dropout 0 the two Q|K|V images are the same numbers: the second training forward reads the first one's image
    (autograd._QKV_IMAGES, xnrs_row_lists::qkv_shared) -- loss and every gradient BITWISE equal to the step that projects
    twice; and no reuse once the input was modified in place or goes through an input dropout."""
    from xnrs_amd import autograd as AG
    c = dict(model="NRMS", B=6, H=5, C=3, S=20, D=64, h=4, E=32, bias=False, seed=777, min_len=2)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    model.train()  # attention dropout 0.1 on: the two encodes draw independently, the projection is shared all the same
    for mod in model.modules():
        if isinstance(mod, layers.MultiHeadAttention):
            mod.dropout.p = 0.0  # (deterministic comparison: same probabilities in both runs)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    labels = torch.tensor([0, 1, 0, 2, 1, 0], device=DEV)

    def step(share):
        old, AG.SHARE_QKV = AG.SHARE_QKV, share
        old_merge, AG.MERGE_DW = AG.MERGE_DW, False  # (the merged dW product changes the summation order: tested in
        old_out, AG.SHARE_OUTPUTS = AG.SHARE_OUTPUTS, False  # tests/test_hip_train_step.py; and with dropout 0 the whole
        try:                                                  # second encode would be shared: here the projection alone)
            model.zero_grad(set_to_none=True)
            before = AG.STATS["shared_qkv_forwards"]
            preds = torch.relu(model(batch))
            ue = model.get_user_embeddings(batch)
            took = AG.STATS["shared_qkv_forwards"] - before
            loss = torch.nn.functional.mse_loss(preds, batch["targets"]) + 0.1 * contrastive_loss(ue, labels, 0.08)
            loss.backward()
        finally:
            AG.SHARE_QKV = old
            AG.MERGE_DW = old_merge
            AG.SHARE_OUTPUTS = old_out
        return loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, took

    l0, g0, t0 = step(False)
    l1, g1, t1 = step(True)
    assert t0 == 0 and t1 == 1  # the second history encode (the candidates and the user tower see other inputs)
    assert torch.equal(l0, l1) and g0.keys() == g1.keys()
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # an in-place change of the input between the two encodes: another version counter, no reuse
    AG.SHARE_OUTPUTS = False
    hx, hm = batch["user_features"]["history"]["title_emb"]
    before = AG.STATS["shared_qkv_forwards"]
    preds = model(batch)
    hx.mul_(1.0)
    model.get_user_embeddings(batch)
    assert AG.STATS["shared_qkv_forwards"] == before
    del preds
    # input dropout > 0 (no shipped config): every encode sees its own dropped copy, no reuse
    model.news_encoder.dropout.p = 0.5
    before = AG.STATS["shared_qkv_forwards"]
    preds = model(batch)
    model.get_user_embeddings(batch)
    assert AG.STATS["shared_qkv_forwards"] == before
    model.news_encoder.dropout.p = 0.0
    AG.SHARE_OUTPUTS = True


@pytest.mark.parametrize("n_rows,K,shape", [(20, 100, (64, 25)), (301, 100, (7, 3)), (5, 36, (1, 1)), (33, 600, (9, 11)), (20, 512, (64, 30))])
def test_embedding_table_gradient_vs_index_add(n_rows, K, shape):
    """nn.Embedding backward behind the fused embedding + Linear op (naml.py:82-86): the ballot-scan kernel (K <= 512: four
    waves over the quarters of the id list) and the wide fallback against an fp64 index_add of the same row gradients; list
    lengths that are no multiple of 4 or 64, rows nobody refers to (zero gradient), one-element lists."""
    from xnrs_amd import ops
    g = torch.Generator().manual_seed(1000 + n_rows + K)
    emb = torch.nn.Embedding(n_rows, K)
    fc = torch.nn.Linear(K, 24)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(n_rows, K, generator=g))
        fc.weight.copy_(torch.randn(24, K, generator=g) / K ** 0.5)
        fc.bias.copy_(torch.randn(24, generator=g))
    idx = torch.randint(0, n_rows, shape, generator=g)
    if n_rows > 8:
        idx[idx == 3] = 4            # row 3 is never referred to
    dy = torch.randn(*shape, 24, generator=g)
    w64, t64, b64 = fc.weight.detach().double().clone(), emb.weight.detach().double().clone(), fc.bias.detach().double().clone()
    emb_d, fc_d = emb.to(DEV), fc.to(DEV)  # (nn.Module.to moves in place: the fp64 copies above stay on the host)
    y = ops.embedding_linear(idx.to(DEV), emb_d, fc_d)
    y.backward(dy.to(DEV))
    # reference in fp64
    d_rows = dy.double().reshape(-1, 24) @ w64
    ref = torch.zeros(n_rows, K, dtype=torch.float64).index_add_(0, idx.reshape(-1), d_rows)
    got = emb_d.weight.grad.cpu().double()
    scale = max(ref.abs().max().item(), 1e-6)
    assert (got - ref).abs().max().item() <= 5e-6 * scale * max(1.0, (idx.numel() / n_rows) ** 0.5)
    if n_rows > 8:
        assert torch.equal(got[3], torch.zeros(K, dtype=torch.float64))
    assert torch.allclose(y.detach().cpu().double(), t64[idx] @ w64.t() + b64, rtol=1e-5, atol=1e-5)
