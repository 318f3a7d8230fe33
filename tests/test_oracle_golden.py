"""CPU: the oracle (oracle/xnrs_oracle.py) against the golden vectors recorded from the REAL
reference (tests/golden/make_golden.py).  This is the pin that lets the oracle stand in for the
reference on the GPU box, where /root/reference does not exist."""
import json
import os

import pytest
import torch

from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases

TOL = 2e-6  # oracle and reference are both torch-CPU fp32: agree to rounding noise


@pytest.mark.parametrize("name", sorted(cases.BLOCKS))
def test_blocks(name):
    g = H.golden("blocks")
    c = cases.BLOCKS[name]
    x, m, u = cases.block_inputs(c)
    D = c["D"]
    if c["kind"] == "additive":
        sd = H.state_for(H.additive_shapes(D, c["A"]), c["seed"] + 1)
        y, a = O.additive_attention(x, m if c["mask"] else None, sd, return_weights=True)
        H.assert_close(y, g[f"{name}/y"], TOL, name)
        H.assert_close(a, g[f"{name}/a"], TOL, name)
    elif c["kind"] == "mha":
        sd = H.state_for(H.mha_shapes(D), c["seed"] + 1)
        H.assert_close(O.multi_head_attention(x, m if c["mask"] else None, sd, c["h"]), g[f"{name}/y"], TOL, name)
    elif c["kind"] == "mean":
        H.assert_close(O.masked_mean(x, m), g[f"{name}/y"], TOL, name)
    elif c["kind"] == "dot":
        H.assert_close(O.dot_scoring(u, x, c["normalize"]), g[f"{name}/y"], TOL, name)


@pytest.mark.parametrize("name", sorted(cases.ENCODERS))
def test_encoders(name):
    g = H.golden("encoders")
    c = cases.ENCODERS[name]
    x, m = cases.encoder_inputs(c)
    sd = H.state_for(H.encoder_shapes(c), c["seed"] + 1)
    if c["tower"] == "news":
        y, hm = O.text_encoder(x, m, sd, c["h"])
        H.assert_close(y, g[f"{name}/y"], TOL, name)
        assert torch.equal(hm, torch.from_numpy(g[f"{name}/hm"]))
    else:
        if c["pooler"] == "additive":
            y, a = O.user_encoder(x, m, sd, c["h"], return_weights=True)
            H.assert_close(a, g[f"{name}/a"], TOL, name)
        else:
            y = O.user_encoder(x, m, sd, c["h"])
        H.assert_close(y, g[f"{name}/y"], TOL, name)


@pytest.mark.parametrize("name", sorted(cases.MODELS))
def test_models(name):
    g = H.golden("models")
    c = cases.MODELS[name]
    sd = H.state_for(H.model_shapes(c), c["seed"] + 1)
    batch = cases.model_batch(c)
    if c["model"] == "NAML":
        H.assert_close(O.naml_forward(batch, sd), g[f"{name}/r"], TOL, name)
        H.assert_close(O.naml_user_embeddings(batch, sd), g[f"{name}/ue"], TOL, name)
        return
    hist = batch["user_features"]["history"]["title_emb"]
    cand = batch["candidate_features"]["title_emb"]
    r, u, cc = O.parent_forward(hist, cand, sd, c["h"], return_embeddings=True)
    H.assert_close(r, g[f"{name}/r"], TOL, name)
    H.assert_close(u, g[f"{name}/u"], TOL, name)
    H.assert_close(cc, g[f"{name}/c"], TOL, name)
    H.assert_close(O.parent_user_embeddings(hist, sd, c["h"]), g[f"{name}/ue"], TOL, name)


def test_lstur_news_encoder():
    g = H.golden("lstur")
    c = cases.LSTUR
    sd = H.state_for(H.model_shapes(c), c["seed"] + 1)
    x, m, ci, si = cases.lstur_inputs(c)
    e, mm = O.lstur_news_encoder((x, m), ci, si, sd)
    H.assert_close(e, g["lstur_news/e"], TOL)
    assert torch.equal(mm, torch.from_numpy(g["lstur_news/m"]))


def test_infonce_value_and_grad():
    g = H.golden("grads")
    e, lab = cases.infonce_inputs()
    e.requires_grad_(True)
    l = O.contrastive_loss(e, lab, 0.08)
    l.backward()
    H.assert_close(l, g["infonce/loss"], 1e-5)
    H.assert_close(e.grad, g["infonce/grad"], 1e-5)
    l2 = O.contrastive_loss_loop(e.detach(), lab, 0.08)
    H.assert_close(l2, g["infonce/loss"], 1e-5)


def test_train_step_loss_and_grads():
    g = H.golden("grads")
    c = cases.GRAD
    sd = H.state_for(H.model_shapes(c), c["seed"] + 1)
    sd = {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in sd.items()}
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx.requires_grad_(True)
    cx.requires_grad_(True)
    labels = cases.theme_labels(batch["main_theme"])
    loss, lrec, lcl = O.train_step_loss(batch, sd, c["h"], labels, c["temperature"], c["lambda_cl"])
    loss.backward()
    H.assert_close(loss, g["grad/loss"], 1e-5)
    H.assert_close(lrec, g["grad/loss_rec"], 1e-5)
    H.assert_close(lcl, g["grad/loss_cl"], 1e-5)
    H.assert_close(hx.grad, g["grad/d_hist_x"], 2e-5)
    H.assert_close(cx.grad, g["grad/d_cand_x"], 2e-5)
    n = H.assert_grads_close({k: v.grad for k, v in sd.items() if v.grad is not None}, g, 5e-5)
    assert n >= 28


def test_train_step_at_the_shipped_shape():
    """The oracle's train step at S=50, D=768, 16 heads, E=256 (cases.GRAD_SHIPPED) against the real reference's loss and
    (sampled) gradients -- the checker of tests/test_hip_grads.py::test_train_step_matches_reference_golden_shipped_shape."""
    g = H.golden("grads_shipped")
    c = cases.GRAD_SHIPPED
    sd = H.state_for(H.model_shapes(c), c["seed"] + 1)
    sd = {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in sd.items()}
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx.requires_grad_(True)
    cx.requires_grad_(True)
    labels = cases.theme_labels(c["themes"])
    loss, lrec, lcl = O.train_step_loss(batch, sd, c["h"], labels, c["temperature"], c["lambda_cl"])
    loss.backward()
    assert float(g["gs/loss_cl"]) > 0.0  # the contrastive term is live (two of the three impressions share a theme)
    H.assert_close(loss, g["gs/loss"], 1e-5)
    H.assert_close(lrec, g["gs/loss_rec"], 1e-5)
    H.assert_close(lcl, g["gs/loss_cl"], 1e-5)
    assert cases.grad_sample_idx(hx.grad.numel()) is not None and g["gs/d_hist_x"].shape == (cases.GRAD_SAMPLE_N,)
    H.assert_close(cases.grad_sample(hx.grad), g["gs/d_hist_x"], 5e-5)
    H.assert_close(cases.grad_sample(cx.grad), g["gs/d_cand_x"], 5e-5)
    n = H.assert_sampled_grads_close({k: v.grad for k, v in sd.items() if v.grad is not None}, g, 5e-5)
    assert n >= 28


def test_standardrec_train_step_at_the_shipped_shape():
    """The same for the attention-free bi-encoder of BASELINE configs[3] (StandardRec, biases on; cases.GRAD_SHIPPED_STD):
    the oracle against the real reference's loss and (sampled) gradients -- the checker of
    tests/test_hip_grads.py::test_standardrec_train_step_matches_reference_golden_shipped_shape."""
    g = H.golden("grads_shipped_standard")
    c = cases.GRAD_SHIPPED_STD
    sd = H.state_for(H.model_shapes(c), c["seed"] + 1)
    sd = {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in sd.items()}
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx.requires_grad_(True)
    cx.requires_grad_(True)
    labels = cases.theme_labels(c["themes"])
    loss, lrec, lcl = O.train_step_loss(batch, sd, None, labels, c["temperature"], c["lambda_cl"])
    loss.backward()
    assert float(g["gss/loss_cl"]) > 0.0
    H.assert_close(loss, g["gss/loss"], 1e-5)
    H.assert_close(lrec, g["gss/loss_rec"], 1e-5)
    H.assert_close(lcl, g["gss/loss_cl"], 1e-5)
    H.assert_close(cases.grad_sample(hx.grad), g["gss/d_hist_x"], 5e-5)
    H.assert_close(cases.grad_sample(cx.grad), g["gss/d_cand_x"], 5e-5)
    n = H.assert_sampled_grads_close({k: v.grad for k, v in sd.items() if v.grad is not None}, g, 5e-5, "gss/dW/", "gss/max/")
    assert n >= 16


def test_naml_train_step_at_the_shipped_shape():
    """... and for BASELINE configs[4]'s model (NAML: two additive text views, category / subcategory embeddings through a
    Linear, feature pooler; cases.GRAD_SHIPPED_NAML): the oracle against the real reference's loss and (sampled) gradients
    incl. the embedding tables -- the checker of tests/test_hip_grads.py::test_naml_train_step_matches_reference_golden_shipped_shape."""
    g = H.golden("grads_shipped_naml")
    c = cases.GRAD_SHIPPED_NAML
    sd = H.state_for(H.model_shapes(c), c["seed"] + 1)
    sd = {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in sd.items()}
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx.requires_grad_(True)
    cx.requires_grad_(True)
    labels = cases.theme_labels(c["themes"])
    lrec = O.mse_relu_loss(O.naml_forward(batch, sd), batch["targets"])
    ue = O.naml_user_embeddings(batch, sd)
    lcl = O.contrastive_loss(ue.reshape(ue.shape[0], -1), labels, c["temperature"])
    loss = lrec + c["lambda_cl"] * lcl
    loss.backward()
    assert float(g["gsn/loss_cl"]) > 0.0
    H.assert_close(loss, g["gsn/loss"], 1e-5)
    H.assert_close(lrec, g["gsn/loss_rec"], 1e-5)
    H.assert_close(lcl, g["gsn/loss_cl"], 1e-5)
    H.assert_close(cases.grad_sample(hx.grad), g["gsn/d_hist_x"], 5e-5)
    H.assert_close(cases.grad_sample(cx.grad), g["gsn/d_cand_x"], 5e-5)
    # (2e-4, the GPU tests' bar, instead of 5e-5: the fc2 BIAS gradients are sums that cancel analytically -- sum_i de_i =
    # (1 - sum a) c ~ 1e-8 c -- so what two fp32 implementations return for them is rounding noise of the terms; the
    # feature pooler's lands at 1e-4 of the floor scale between the oracle and the reference, both on the CPU)
    n = H.assert_sampled_grads_close({k: v.grad for k, v in sd.items() if v.grad is not None}, g, 2e-4, "gsn/dW/", "gsn/max/")
    assert n >= 30


def test_quirks_pinned():
    """The parity-critical quirks of SURVEY.md finding 4, checked on the oracle."""
    torch.manual_seed(0)
    D, h, S = 32, 4, 8
    sd = H.state_for(H.mha_shapes(D), 7)
    x = torch.randn(1, S, D)
    m = torch.ones(1, S, 1)
    m[0, 5:] = 0
    y0 = O.multi_head_attention(x, m, sd, h)
    x2 = x.clone()
    x2[0, 5:] += 1.0  # perturb ONLY masked token positions
    y1 = O.multi_head_attention(x2, m, sd, h)
    assert (y0[0, :5] - y1[0, :5]).abs().max() > 1e-3, "row mask: masked keys must still influence valid rows"
    # additive attention: all-masked -> weights exactly 0 -> pooled vector exactly 0
    asd = H.state_for(H.additive_shapes(D, 256), 8)
    out, a = O.additive_attention(x, torch.zeros(1, S, 1), asd, return_weights=True)
    assert a.abs().max() == 0 and out.abs().max() == 0
    # head-bias leak: fully padded news -> y = W4 relu(b3) + b4 != 0
    c = cases.ENCODERS["news_nrms_tiny"]
    esd = H.state_for(H.encoder_shapes(c), 9)
    xx = torch.zeros(1, 1, c["S"], c["D"])
    y, hm = O.text_encoder(xx, torch.zeros(1, 1, c["S"], 1), esd, c["h"])
    assert y.abs().max() > 1e-3 and hm.item() == 0.0


def test_d_mod_h_error_matches_reference():
    meta = json.load(open(os.path.join(H.GOLDEN_DIR, "meta.json")))
    assert meta["errors"]["err"] == "RuntimeError"
    sd = H.state_for(H.mha_shapes(300), 1)
    with pytest.raises(RuntimeError) as ei:
        O.multi_head_attention(torch.zeros(2, 30, 300), None, sd, 16)
    assert str(ei.value) == meta["errors"]["msg"]


def test_out_projection_behind_the_pooling_is_exact_algebra():
    """The identity the HIP path's fold rests on, checked on the oracle in float64:
    pooler(att(x)) with Y = O Wo^T + bo per token  ==  Wo (sum_i a_i O_i) + bo (sum_i a_i), scores from (W1 Wo) O_i + (W1 bo + b1)
    -- for masks with holes, an all-masked and a fully live sequence."""
    from xnrs_amd import synth
    torch.manual_seed(0)
    B, S, D, h, A = 5, 13, 24, 4, 10
    shapes = {"q_linear.weight": (D, D), "q_linear.bias": (D,), "k_linear.weight": (D, D), "k_linear.bias": (D,),
              "v_linear.weight": (D, D), "v_linear.bias": (D,), "out.weight": (D, D), "out.bias": (D,),
              "fc1.weight": (A, D), "fc1.bias": (A,), "fc2.weight": (1, A), "fc2.bias": (1,)}
    sd = {k: v.double() for k, v in synth.fill_state_dict(shapes, 901).items()}
    x = torch.randn(B, S, D, dtype=torch.float64)
    m = (torch.rand(B, S, 1) < 0.6).double()
    m[0] = 0
    m[1] = 1
    ref = O.additive_attention(O.multi_head_attention(x, m, sd, h), m, sd)  # the reference's order
    # the attention rows BEFORE the out-projection: run the block with an identity out layer
    eye = dict(sd, **{"out.weight": torch.eye(D, dtype=torch.float64), "out.bias": torch.zeros(D, dtype=torch.float64)})
    o = O.multi_head_attention(x, m, eye, h)
    wf = sd["fc1.weight"] @ sd["out.weight"]
    bf = sd["fc1.weight"] @ sd["out.bias"] + sd["fc1.bias"]
    po, a = O.additive_attention(o, m, {"fc1.weight": wf, "fc1.bias": bf, "fc2.weight": sd["fc2.weight"],
                                          "fc2.bias": sd["fc2.bias"]}, return_weights=True)
    folded = po @ sd["out.weight"].t() + a.sum(dim=1, keepdim=True) * sd["out.bias"]
    assert (folded - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
    assert folded[0].abs().max().item() == 0.0  # all-masked: exactly zero either way
