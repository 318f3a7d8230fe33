"""GPU: integrated gradients (xnrs/explain.py:144-182) through the HIP input-gradient path -- the batched form of
xnrs_amd.explain (all interpolation steps as ONE batch) against the reference's loop restated on the CPU oracle, against
the library's own loop form, and the bookkeeping that makes it cheap (no parameter gradient is computed in such a pass)."""
import pytest
import torch

from oracle import xnrs_oracle as O
from tests.golden import cases
from tests.test_hip_grads import Cfg, load
from xnrs_amd import autograd as AG, synth
from xnrs_amd.explain import integrated_gradients
from xnrs_amd.models import make_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_ig(sd, n_heads, hist, cand, cidx, n_steps):
    """explain_score_in_batch's loop (explain.py:152-173) over the CPU oracle, fp32, torch autograd."""
    hx, hm = hist
    nsd = {k[len("news_encoder."):]: v for k, v in sd.items() if k.startswith("news_encoder.")}
    usd = {k[len("user_encoder."):]: v for k, v in sd.items() if k.startswith("user_encoder.")}
    c, _ = O.text_encoder(cand[0][:, cidx:cidx + 1], cand[1][:, cidx:cidx + 1], nsd, n_heads)
    da = 1.0 / n_steps
    grads = []
    sa = None
    for a in torch.arange(da, 1 + da, da)[:n_steps]:
        ga = (a * hx).requires_grad_()
        ha, ham = O.text_encoder(ga, hm, nsd, n_heads)
        ua = O.user_encoder(ha, ham, usd, n_heads)
        sa = torch.relu(O.dot_scoring(ua, c.detach()))
        grads.append(torch.autograd.grad(sa, ga)[0])
    int_grads = torch.sum(torch.cat(grads) * da, dim=0)
    attr = (int_grads * hx[0]).sum(dim=2)
    return attr, int_grads, float(sa.item())


@pytest.mark.parametrize("name", ["NRMS", "standard"])
def test_batched_ig_equals_the_reference_loop_on_the_oracle(name):
    c = dict(model=name, B=1, H=7, C=3, S=12, D=64, h=4, E=32, bias=True, seed=5101, min_len=4)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)  # eval mode: explain.py loads the checkpoint and calls .eval()
    batch = cases.model_batch(c)
    hist = batch["user_features"]["history"]["title_emb"]
    cand = batch["candidate_features"]["title_emb"]
    n_steps = 24
    with torch.no_grad():  # explain the best-scored candidate: a score the ReLU has cut to 0 has no gradient to compare
        cidx = int(O.parent_forward(hist, cand, sd, c["h"]).reshape(-1).argmax())
    attr_o, ig_o, s_o = oracle_ig(sd, c["h"], hist, cand, cidx, n_steps)
    d = lambda t: t.to(DEV)
    out = integrated_gradients(model, d(hist[0]), d(hist[1]), d(cand[0]), d(cand[1]), candidate_idx=cidx, n_steps=n_steps)
    scale = ig_o.abs().max().item()
    assert scale > 0  # (a dead score would make the comparison vacuous)
    assert (out["int_grads"].cpu() - ig_o).abs().max().item() <= 2e-5 * scale
    assert (out["attr"].cpu() - attr_o).abs().max().item() <= 2e-5 * attr_o.abs().max().item()
    assert abs(out["s_true"] - s_o) <= 2e-5 * max(abs(s_o), 1e-3)
    # the library's own loop form (the reference's order of calls) and chunked batches give the same numbers
    loop = integrated_gradients(model, d(hist[0]), d(hist[1]), d(cand[0]), d(cand[1]), candidate_idx=cidx, n_steps=n_steps, batched=False)
    chunk = integrated_gradients(model, d(hist[0]), d(hist[1]), d(cand[0]), d(cand[1]), candidate_idx=cidx, n_steps=n_steps, steps_per_batch=5)
    for other in (loop, chunk):
        assert (other["int_grads"] - out["int_grads"]).abs().max().item() <= 2e-6 * scale
        assert abs(other["s_true"] - out["s_true"]) <= 1e-6 * max(abs(s_o), 1e-3)
    assert all(p.grad is None for p in model.parameters())  # an explanation leaves the parameters alone


def test_an_input_gradient_pass_computes_no_parameter_gradient(monkeypatch):
    """torch.autograd.grad(score, tokens): the backward nodes ask the engine which of their inputs' gradients this pass
    uses (autograd._wanted_inputs) and skip every weight-gradient product; a plain backward() still fills every .grad."""
    c = dict(model="NRMS", B=2, H=5, C=2, S=10, D=64, h=4, E=32, bias=True, seed=5202, min_len=3)
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    seen = []
    real = AG._wanted_inputs

    def spy(ctx, is_tensor, first):
        w = real(ctx, is_tensor, first)
        seen.append([v for v, t in zip(w, is_tensor[first:]) if t])  # (the parameters that exist)
        return w
    monkeypatch.setattr(AG, "_wanted_inputs", spy)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    hx = hx.clone().requires_grad_()
    batch["user_features"]["history"]["title_emb"] = (hx, hm)
    r = torch.relu(model(batch)).sum()
    (g,) = torch.autograd.grad(r, hx, retain_graph=True)
    assert seen and not any(any(w) for w in seen)        # no node was asked for a parameter gradient
    assert torch.isfinite(g).all() and g.abs().max() > 0
    seen.clear()
    r.backward()
    assert seen and all(w and all(w) for w in seen)      # the plain backward wants every parameter there is
    g2 = hx.grad
    assert torch.equal(g, g2)                            # and the input gradient is the same bits either way
    assert all(p.grad is not None for n, p in model.named_parameters() if "dummy" not in n)
