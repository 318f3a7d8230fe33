"""GPU: seeded random sweep over shapes and module options -- the HIP path (every kernel-selection branch: generic /
head-per-wave / LDS-staged attention, vector and scalar GEMM loads, ragged tiles, K tails) against the CPU oracle,
plus the exact switches (skip_empty, unpadded) and the bf16x3 GEMM mode on the same inputs.

Tolerance: the 1e-4 bar of BASELINE.json (tests/helpers.py RTOL)."""
import os

import numpy as np
import pytest
import torch

from oracle import xnrs_oracle as O
from tests import helpers as H
from xnrs_amd import hip, synth
from xnrs_amd.models.components import layers, news_encoding, user_encoding, scoring, ParentRec

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(seed):
    rng = np.random.default_rng(1000 + seed)
    h = int(rng.choice([1, 2, 3, 4, 5, 8]))
    dk = int(rng.choice([2, 4, 6, 8, 12, 16, 20, 48, 64, 72]))   # d_k % 4 != 0 -> scalar attention path; > 64 -> generic
    D = h * dk
    S = int(rng.choice([1, 2, 7, 16, 17, 30, 32, 33, 50, 64, 65, 100]))
    if D > 384:
        S = min(S, 33)
    return dict(h=h, D=D, S=S, E=int(rng.choice([8, 24, 64])), A=int(rng.choice([16, 100, 256])),
                B=int(rng.integers(1, 4)), Hn=int(rng.integers(1, 7)), C=int(rng.integers(1, 5)),
                att_news=bool(rng.random() < 0.75), att_user=bool(rng.random() < 0.6),
                pool_news="additive" if rng.random() < 0.75 else "mean", pool_user="additive" if rng.random() < 0.7 else "mean",
                head_news=bool(rng.random() < 0.7), head_user=bool(rng.random() < 0.4), bias=bool(rng.random() < 0.6),
                holes=bool(rng.random() < 0.4), seed=seed)


def _pool(kind, D, A):
    return layers.AdditiveAttention(D, A) if kind == "additive" else layers.MaskedMean()


@pytest.mark.parametrize("seed", range(28))
def test_random_bi_encoder_matches_oracle(seed):
    c = _cfg(seed)
    D, S, E, A, h = c["D"], c["S"], c["E"], c["A"], c["h"]
    Eo = E if c["head_news"] else D          # width of the news vectors (user tower width)
    hu = next(k for k in (h, 4, 3, 2, 1) if Eo % k == 0)
    news = news_encoding.TextEncoder(pooler=_pool(c["pool_news"], D, A), p_dropout=0.0, out_features=Eo, in_features=D,
                                     head=c["head_news"], att=layers.MultiHeadAttention(h, D) if c["att_news"] else None,
                                     bias=c["bias"])
    user = user_encoding.UserEncoder(pooler=_pool(c["pool_user"], Eo, A), p_dropout=0.0, emb_dim=Eo,
                                     att=layers.MultiHeadAttention(hu, Eo) if c["att_user"] else None,
                                     head=c["head_user"], bias=c["bias"])
    model = ParentRec(news, user, scoring.DotScoring())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 7000 + seed)
    model.load_state_dict(sd)
    model = model.eval().to(DEV)

    rng = synth.rng_for(8000 + seed)
    n_hist = rng.integers(0, c["Hn"] + 1, size=(c["B"],))       # 0 = a user without history
    hx, hm = synth.token_block(rng, c["B"], c["Hn"], S, D, min_len=1, n_valid=n_hist, full_pad_prob=0.15)
    cx, cm = synth.token_block(rng, c["B"], c["C"], S, D, min_len=1)
    if c["holes"]:
        hm = hm * torch.from_numpy((rng.random(tuple(hm.shape)) < 0.7).astype("float32"))
        cm = cm * torch.from_numpy((rng.random(tuple(cm.shape)) < 0.7).astype("float32"))
    hist, cand = (hx, hm), (cx, cm)
    # the oracle needs per-tower head counts: assemble it from the tower functions
    nsd = {k[len("news_encoder."):]: v for k, v in sd.items() if k.startswith("news_encoder.")}
    usd = {k[len("user_encoder."):]: v for k, v in sd.items() if k.startswith("user_encoder.")}
    hv, hmask = O.text_encoder(hx, hm, nsd, h if c["att_news"] else None)
    cv, _ = O.text_encoder(cx, cm, nsd, h if c["att_news"] else None)
    u = O.user_encoder(hv, hmask, usd, hu if c["att_user"] else None)
    ref = O.dot_scoring(u, cv)

    def run():
        with torch.no_grad():
            return model._forward(hist, cand, return_embeddings=True)

    what = f"seed {seed}: {c}"
    r, uu, cc = run()
    H.assert_close(cc, cv, what=what + " cand")
    H.assert_close(uu, u, what=what + " user")
    H.assert_close(r, ref, what=what + " scores")
    # the exact switches
    for flags in (dict(skip_empty=True), dict(unpadded=True), dict(skip_empty=True, unpadded=True)):
        for k, v in flags.items():
            setattr(model.news_encoder, k, v)
        try:
            r2, u2, c2 = run()
        finally:
            model.news_encoder.skip_empty = model.news_encoder.unpadded = False
        H.assert_close(c2, cc, tol=2e-6, what=what + f" {flags} cand")
        H.assert_close(r2, r, tol=2e-5, what=what + f" {flags} scores")
    # the fp32-grade split mode on every forward GEMM
    prev = hip.set_gemm_mode(hip.GEMM_BF16X3)
    os.environ["XNRS_GEMM_SPLIT_MIN_TILES"] = "0"
    hip.reload_knobs()
    try:
        r3, _, c3 = run()
    finally:
        hip.set_gemm_mode(prev)
        os.environ.pop("XNRS_GEMM_SPLIT_MIN_TILES", None)
        hip.reload_knobs()
    H.assert_close(c3, cv, what=what + " bf16x3 cand")
    H.assert_close(r3, ref, what=what + " bf16x3 scores")


# seeds 200..231: the round-3 soak range (tools/soak_random_shapes.py), pinned in round 4 together with the bar below
@pytest.mark.parametrize("seed", list(range(16)) + list(range(200, 232)))
def test_random_bi_encoder_gradients_match_oracle(seed):
    """The hand-written backward (xnrs_seq_encoder_bwd and friends) against torch autograd of the CPU oracle on
    the same random configurations: parameter gradients of both towers and the input gradients d loss / d x that
    the explainer needs (explain.py:160-166).

    `pooler.fc2.bias` of an additive pooler has its own bar.  Its gradient is sum_i de_i with de_i = a_i (da_i - sum_j a_j da_j):
    per sequence that is (sum_i a_i da_i) (1 - sum_i a_i) = O(1e-8) -- the bias shifts every score of a sequence alike and only
    the `+ 1e-8` of layers.py:64 keeps the weights from being invariant to it.  Both sides therefore compare two fp32 sums of
    n = rows terms that cancel analytically; what is left on EITHER side is summation noise of order sqrt(n) * 2^-23 * |de|,
    and |de| is bounded through its sibling gradient d fc2.weight = sum_i de_i tanh(.)_i.  Bar for this key: the usual
    2e-4 of the scale PLUS 4 sqrt(n) 2^-23 max|d fc2.weight| (one round-3 soak seed sat at 2.1e-4 of the scale without it)
    PLUS 1e-6 of the largest gradient of the step: each de_i is itself the difference a_i da_i - a_i c of two products that
    can be far larger than de_i (c = sum_j a_j da_j contains the whole shift of the scores), so the noise that survives the
    cancellation is a few ulps of THOSE, on both sides (round-4 soak seed 354: 2.6e-7 of the largest gradient)."""
    c = _cfg(100 + seed)
    D, S, A, h = c["D"], min(c["S"], 64), c["A"], c["h"]
    Eo = c["E"] if c["head_news"] else D
    hu = next(k for k in (h, 4, 3, 2, 1) if Eo % k == 0)
    news = news_encoding.TextEncoder(pooler=_pool(c["pool_news"], D, A), p_dropout=0.0, out_features=Eo, in_features=D,
                                     head=c["head_news"], att=layers.MultiHeadAttention(h, D) if c["att_news"] else None,
                                     bias=c["bias"])
    user = user_encoding.UserEncoder(pooler=_pool(c["pool_user"], Eo, A), p_dropout=0.0, emb_dim=Eo,
                                     att=layers.MultiHeadAttention(hu, Eo) if c["att_user"] else None,
                                     head=c["head_user"], bias=c["bias"])
    model = ParentRec(news, user, scoring.DotScoring())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 7100 + seed)
    model.load_state_dict(sd)
    model = model.eval().to(DEV)   # eval: dropout off, gradients on

    rng = synth.rng_for(8100 + seed)
    Hn = max(c["Hn"], 2)
    n_hist = rng.integers(1, Hn + 1, size=(c["B"],))
    hx, hm = synth.token_block(rng, c["B"], Hn, S, D, min_len=1, n_valid=n_hist)
    cx, cm = synth.token_block(rng, c["B"], c["C"], S, D, min_len=1)
    wgt = torch.from_numpy(rng.standard_normal((c["B"], c["C"], 1)).astype("float32"))

    hxd, cxd = hx.to(DEV).requires_grad_(True), cx.to(DEV).requires_grad_(True)
    r = model._forward((hxd, hm), (cxd, cm))
    (r * wgt.to(DEV)).sum().backward()

    osd = {k: v.clone().requires_grad_(not k.endswith("dummy_param")) for k, v in sd.items()}
    nsd = {k[len("news_encoder."):]: v for k, v in osd.items() if k.startswith("news_encoder.")}
    usd = {k[len("user_encoder."):]: v for k, v in osd.items() if k.startswith("user_encoder.")}
    hxo, cxo = hx.clone().requires_grad_(True), cx.clone().requires_grad_(True)
    hv, hmask = O.text_encoder(hxo, hm, nsd, h if c["att_news"] else None)
    cv, _ = O.text_encoder(cxo, cm, nsd, h if c["att_news"] else None)
    ro = O.dot_scoring(O.user_encoder(hv, hmask, usd, hu if c["att_user"] else None), cv)
    (ro * wgt).sum().backward()

    what = f"seed {seed}: {c}"
    H.assert_close(r, ro, what=what + " fwd")
    gmax = max(v.grad.abs().max().item() for v in osd.values() if v.grad is not None)
    assert gmax > 0
    H.assert_close(hxd.grad, hxo.grad, 2e-4, what + " d hist x")
    H.assert_close(cxd.grad, cxo.grad, 2e-4, what + " d cand x")
    n = 0
    for k, p in model.named_parameters():
        ref = osd[k].grad
        if ref is None:
            continue
        assert p.grad is not None, f"{what}: no grad for {k}"
        scale = max(ref.abs().max().item(), 1e-3 * gmax)
        e = (p.grad.cpu().double() - ref.double()).abs().max().item() / scale
        bar = 2e-4
        if k.endswith("pooler.fc2.bias"):
            rows = (c["B"] * Hn * S) if k.startswith("news_encoder.") else c["B"] * Hn
            if k.startswith("news_encoder."):
                rows += c["B"] * c["C"] * S  # the candidates go through the same tower
            gw = osd[k[:-len("bias")] + "weight"].grad.abs().max().item()
            bar += 4.0 * (rows ** 0.5) * 2.0 ** -23 * gw / scale + 1e-6 * gmax / scale
        assert e <= bar, f"{what}: {k}: {e:.3e} (bar {bar:.3e})"
        n += 1
    assert n >= 2
