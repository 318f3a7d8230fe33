"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs and against the committed golden vectors recorded from the real reference.

Tolerance: BASELINE.json north_star -- fp32 scores within 1e-4 relative of the reference CPU path
(tests/helpers.py RTOL; "relative" = max|d| / max|ref| per tensor)."""
import numpy as np
import pytest
import torch

from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases
from xnrs_amd import hip, synth
from xnrs_amd.models import make_model
from xnrs_amd.models.components import layers, news_encoding, user_encoding, scoring

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg(dict):
    __getattr__ = dict.__getitem__


def load(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed)
    module.load_state_dict(sd)
    module.eval()
    return module.to(DEV), sd


@pytest.mark.parametrize("name", sorted(cases.BLOCKS))
def test_blocks(name):
    g = H.golden("blocks")
    c = cases.BLOCKS[name]
    x, m, u = cases.block_inputs(c)
    D = c["D"]
    with torch.no_grad():
        if c["kind"] == "additive":
            mod, sd = load(layers.AdditiveAttention(D, c["A"]), c["seed"] + 1)
            y, a = mod(x.to(DEV), m.to(DEV) if c["mask"] else None, return_weights=True)
            yo, ao = O.additive_attention(x, m if c["mask"] else None, sd, return_weights=True)
            H.assert_close(y, yo, what=name + " vs oracle")
            H.assert_close(a, ao, what=name + " weights vs oracle")
            H.assert_close(y, g[f"{name}/y"], what=name + " vs golden")
            H.assert_close(a, g[f"{name}/a"], what=name + " weights vs golden")
            y1 = mod(x.to(DEV), m.to(DEV) if c["mask"] else None)
            assert torch.equal(y1, y)
        elif c["kind"] == "mha":
            mod, sd = load(layers.MultiHeadAttention(c["h"], D), c["seed"] + 1)
            y = mod(x.to(DEV), m.to(DEV) if c["mask"] else None)
            H.assert_close(y, O.multi_head_attention(x, m if c["mask"] else None, sd, c["h"]), what=name + " vs oracle")
            H.assert_close(y, g[f"{name}/y"], what=name + " vs golden")
        elif c["kind"] == "mean":
            y = layers.MaskedMean()(x.to(DEV), m.to(DEV))
            H.assert_close(y, g[f"{name}/y"], what=name)
        elif c["kind"] == "dot":
            y = scoring.DotScoring(normalize=c["normalize"])(u.to(DEV), x.to(DEV))
            H.assert_close(y, g[f"{name}/y"], what=name)


@pytest.mark.parametrize("name", sorted(cases.ENCODERS))
def test_encoders(name):
    g = H.golden("encoders")
    c = cases.ENCODERS[name]
    x, m = cases.encoder_inputs(c)
    D, E, A = c["D"], c["E"], c["A"]
    att = layers.MultiHeadAttention(c["h"], D) if c["att"] else None
    pooler = layers.AdditiveAttention(D, A) if c["pooler"] == "additive" else layers.MaskedMean()
    with torch.no_grad():
        if c["tower"] == "news":
            enc, sd = load(news_encoding.TextEncoder(pooler=pooler, p_dropout=0.0, out_features=E, in_features=D,
                                                     head=c["head"], att=att, bias=c["bias"]), c["seed"] + 1)
            y, hm = enc((x, m))  # CPU inputs: the encoder moves them itself like the reference
            assert y.is_cuda and y.shape == (c["B"], c["N"], E) and hm.shape == (c["B"], c["N"], 1)
            H.assert_close(y, g[f"{name}/y"], what=name)
            assert torch.equal(hm.cpu(), torch.from_numpy(g[f"{name}/hm"]))
        else:
            enc, sd = load(user_encoding.UserEncoder(pooler=pooler, p_dropout=0.0, emb_dim=D, att=att,
                                                     head=c["head"], bias=c["bias"]), c["seed"] + 1)
            if c["pooler"] == "additive":
                y, a = enc((x, m), None, return_weights=True)
                H.assert_close(a, g[f"{name}/a"], what=name)
            else:
                y = enc((x, m))
            H.assert_close(y, g[f"{name}/y"], what=name)


@pytest.mark.parametrize("name", sorted(cases.MODELS))
def test_models(name):
    g = H.golden("models")
    c = cases.MODELS[name]
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    assert sorted(model.state_dict().keys()) == sorted(H.model_shapes(c).keys())
    batch = cases.model_batch(c)
    with torch.no_grad():
        if c["model"] == "NAML":
            H.assert_close(model(batch), g[f"{name}/r"], what=name)
            ue = model.get_user_embeddings(batch)
            assert ue.dim() == 3
            H.assert_close(ue, g[f"{name}/ue"], what=name)
            return
        r, u, cc = model(batch, return_embeddings=True)
        H.assert_close(r, g[f"{name}/r"], what=name + " scores")
        H.assert_close(u, g[f"{name}/u"], what=name + " user")
        H.assert_close(cc, g[f"{name}/c"], what=name + " cand")
        H.assert_close(model.get_user_embeddings(batch), g[f"{name}/ue"], what=name + " ue")
        # oracle on the same inputs
        hist = batch["user_features"]["history"]["title_emb"]
        cand = batch["candidate_features"]["title_emb"]
        H.assert_close(r, O.parent_forward(hist, cand, sd, c["h"]), what=name + " vs oracle")


def test_lstur_news_encoder():
    from xnrs_amd.models.full_models import LSTURNewsEncoder
    g = H.golden("lstur")
    c = cases.LSTUR
    cfg = Cfg(cases.model_cfg(c))
    cfg["catg_features"] = ["category_index", "subcategory_index"]
    enc, sd = load(LSTURNewsEncoder(cfg), c["seed"] + 1)
    x, m, ci, si = cases.lstur_inputs(c)
    with torch.no_grad():
        e, mm = enc((x, m), ci, si)
    H.assert_close(e, g["lstur_news/e"])
    assert torch.equal(mm.cpu(), torch.from_numpy(g["lstur_news/m"]))


def test_quirks_on_device():
    """Row mask, a==0 on all-masked, head-bias leak (SURVEY.md finding 4) on the HIP path."""
    D, h, S = 32, 4, 8
    att, sd = load(layers.MultiHeadAttention(h, D), 7)
    torch.manual_seed(0)
    x = torch.randn(1, S, D)
    m = torch.ones(1, S, 1)
    m[0, 5:] = 0
    with torch.no_grad():
        y0 = att(x.to(DEV), m.to(DEV))
        x2 = x.clone()
        x2[0, 5:] += 1.0
        y1 = att(x2.to(DEV), m.to(DEV))
        assert (y0[0, :5] - y1[0, :5]).abs().max() > 1e-3
        H.assert_close(y1, O.multi_head_attention(x2, m, sd, h))
        pool, psd = load(layers.AdditiveAttention(D, 256), 8)
        out, a = pool(x.to(DEV), torch.zeros(1, S, 1, device=DEV), return_weights=True)
        assert a.abs().max() == 0 and out.abs().max() == 0
        c = cases.ENCODERS["news_nrms_tiny"]
        enc, esd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(c["D"], 256), p_dropout=0.0,
                                                  out_features=c["E"], in_features=c["D"],
                                                  att=layers.MultiHeadAttention(c["h"], c["D"])), 9)
        y, hm = enc((torch.zeros(1, 1, c["S"], c["D"]), torch.zeros(1, 1, c["S"], 1)))
        yo, _ = O.text_encoder(torch.zeros(1, 1, c["S"], c["D"]), torch.zeros(1, 1, c["S"], 1), esd, c["h"])
        assert y.abs().max() > 1e-3 and hm.item() == 0.0
        H.assert_close(y, yo)


def test_d_mod_h_raises_like_reference():
    att = layers.MultiHeadAttention(16, 300).to(DEV)
    with torch.no_grad(), pytest.raises(RuntimeError):
        att(torch.zeros(2, 30, 300, device=DEV), None)


def test_gather_matches_materialised():
    """forward_ids(table, ids) == forward((table[ids], mask[ids]))  (SURVEY.md section 8 a0)."""
    c = cases.ENCODERS["news_nrms_300"]
    D, E, S = c["D"], c["E"], c["S"]
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 256), p_dropout=0.0, out_features=E,
                                             in_features=D, att=layers.MultiHeadAttention(c["h"], D)), 21)
    rng = synth.rng_for(22)
    tx, tm = synth.token_block(rng, 1, 40, S, D, min_len=3)
    tx, tm = tx[0].to(DEV), tm[0].to(DEV)
    ids = torch.from_numpy(rng.integers(0, 40, size=(3, 7)).astype("int64")).to(DEV)
    with torch.no_grad():
        y1, hm1 = enc.forward_ids(tx, tm, ids)
        y2, hm2 = enc((tx[ids], tm[ids]))
    assert torch.equal(y1, y2) and torch.equal(hm1, hm2)
    for att_on in (False,):
        enc2, _ = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 256), p_dropout=0.0,
                                                 out_features=E, in_features=D, att=None), 23)
        with torch.no_grad():
            y1, _ = enc2.forward_ids(tx, tm, ids)
            y2, _ = enc2((tx[ids], tm[ids]))
        assert torch.equal(y1, y2)


def test_empty_and_ragged():
    """Empty batch and S not a multiple of anything convenient."""
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(20, 24), p_dropout=0.0, out_features=12,
                                             in_features=20, att=layers.MultiHeadAttention(5, 20)), 31)
    with torch.no_grad():
        y, hm = enc((torch.zeros(0, 3, 7, 20), torch.zeros(0, 3, 7, 1)))
        assert y.shape == (0, 3, 12) and hm.shape == (0, 3, 1)
        rng = synth.rng_for(32)
        x, m = synth.token_block(rng, 2, 3, 7, 20)
        y, hm = enc((x, m))
        yo, hmo = O.text_encoder(x, m, sd, 5)
        H.assert_close(y, yo)
        assert torch.equal(hm.cpu(), hmo)


def test_cpu_tensors_fail_loudly():
    from xnrs_amd.hip import XnrsHipError
    pool = layers.AdditiveAttention(8, 4)  # left on the CPU
    with torch.no_grad(), pytest.raises(XnrsHipError):
        pool(torch.zeros(1, 2, 8), None)


@pytest.mark.parametrize("with_ids", [False, True])
def test_chunked_passes_equal_single_pass(with_ids):
    """The chunk loop of xnrs_text_encoder_fwd (workspace-bounded passes) must not change a bit."""
    from xnrs_amd import ops
    c = cases.ENCODERS["news_nrms_300"]
    D, E, S = c["D"], c["E"], c["S"]
    enc, _ = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 256), p_dropout=0.0, out_features=E,
                                             in_features=D, att=layers.MultiHeadAttention(c["h"], D)), 81)
    rng = synth.rng_for(82)
    tx, tm = synth.token_block(rng, 1, 23, S, D, min_len=2, full_pad_prob=0.1)
    tx, tm = tx[0].to(DEV), tm[0].to(DEV)
    ids = torch.from_numpy(rng.integers(0, 23, size=(37,)).astype("int32")).to(DEV) if with_ids else None
    with torch.no_grad():
        y0, hm0 = ops.text_encoder(tx, tm, enc, ids=ids, chunk=0)
        for chunk in (1, 5, 16):
            y1, hm1 = ops.text_encoder(tx, tm, enc, ids=ids, chunk=chunk)
            assert torch.equal(y0, y1) and torch.equal(hm0, hm1), chunk


def test_long_sequences_generic_attention_path():
    """S = 100 (> 64): the generic one-wave-per-query-tile kernel; also d_k = 80 (> 64)."""
    for (S, D, h) in ((100, 64, 4), (20, 160, 2)):
        att, sd = load(layers.MultiHeadAttention(h, D), 91)
        rng = synth.rng_for(92)
        x = torch.from_numpy(rng.standard_normal((2, S, D)).astype("float32"))
        m = torch.from_numpy(cases.block_mask(rng, 2, S))
        with torch.no_grad():
            y = att(x.to(DEV), m.to(DEV))
        H.assert_close(y, O.multi_head_attention(x, m, sd, h), what=f"S={S} D={D}")
    att = layers.MultiHeadAttention(2, 8).to(DEV)
    from xnrs_amd.hip import XnrsHipError
    with torch.no_grad(), pytest.raises(XnrsHipError):
        att(torch.zeros(1, 129, 8, device=DEV), None)  # S > 128 is outside the supported range: loud, not wrong


def test_forward_is_hipgraph_capturable():
    """The C ABI promises: no allocation, no host sync, everything on the caller's stream -> a whole
    ParentRec forward can be captured into a hipGraph and replayed (small-batch / eval latency path)."""
    c = cases.MODELS["nrms_300"]
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    hist = batch["user_features"]["history"]["title_emb"]
    cand = batch["candidate_features"]["title_emb"]
    static_h = tuple(t.clone() for t in hist)
    static_c = tuple(t.clone() for t in cand)
    with torch.no_grad():
        ref = model._forward(hist, cand)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                model._forward(static_h, static_c)  # warm-up on the side stream (workspace allocation)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model._forward(static_h, static_c)
        static_h[0].zero_()
        g.replay()
        torch.cuda.synchronize()
        assert not torch.equal(out, ref)  # inputs changed -> the replay really recomputed
        static_h[0].copy_(hist[0])
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_full_size_properties():
    """BASELINE configs[2] at full size (B=512, H=50, C=5, S=50, D=768): size-independent properties.

    * impressions are independent: the B=512 step equals, bit for bit, the same impressions scored in ragged
      sub-batches (different chunking of the 28 160 news, different GEMM tile assignment of every row);
    * permuting the candidates of an impression permutes its scores;
    * a slice of the full-size result matches the CPU oracle within the parity bar."""
    import bench
    dev = torch.device(DEV)
    w = dict(bench.WORKLOAD)
    model, sd = bench.build_model(w, dev)
    hist, cand = bench.make_inputs(w, dev, seed=5)
    with torch.no_grad():
        full = bench.step(model, hist, cand)
        assert full.shape == (w["B"], w["C"], 1) and torch.isfinite(full).all()
        parts, b0 = [], 0
        for nb in (37, 200, 1, 274):
            parts.append(bench.step(model, tuple(t[b0:b0 + nb] for t in hist), tuple(t[b0:b0 + nb] for t in cand)))
            b0 += nb
        assert b0 == w["B"] and torch.equal(torch.cat(parts), full)
        perm = torch.tensor([3, 0, 4, 1, 2], device=dev)
        permuted = bench.step(model, hist, tuple(t[:, perm] for t in cand))
        assert torch.equal(permuted, full[:, perm])
        sl = slice(100, 104)
        ref = O.parent_forward(tuple(t[sl].cpu() for t in hist), tuple(t[sl].cpu() for t in cand), sd, w["h"])
    H.assert_close(full[sl], ref, what="full-size slice vs oracle")


def test_skip_empty_slots_is_exact():
    """TextEncoder.skip_empty: all-masked news (empty history slots, whatever their x) take the constant
    head(0) vector; encoding only the live news must not change a value (benchmark shape, ragged histories)."""
    import bench
    dev = torch.device(DEV)
    w = dict(bench.WORKLOAD, B=48)
    model, _ = bench.build_model(w, dev)
    hist, cand = bench.make_inputs(w, dev, seed=9)
    hx = hist[0].clone()
    b_e = int((hist[1][:, -1].reshape(w["B"], -1).sum(1) == 0).nonzero()[0])  # an impression whose last slot is empty
    hx[b_e, -1] = 7.0  # an all-masked slot with NON-zero tokens: still the constant
    hist = (hx, hist[1])
    with torch.no_grad():
        r0, u0, c0 = model._forward(hist, cand, return_embeddings=True)
        h0, hm0 = model.news_encoder(hist)
        model.news_encoder.skip_empty = True
        try:
            r1, u1, c1 = model._forward(hist, cand, return_embeddings=True)
            h1, hm1 = model.news_encoder(hist)
            all_empty = model.news_encoder((torch.ones_like(hx[:2]), torch.zeros_like(hist[1][:2])))  # no live news at all
            dense_empty = None
        finally:
            model.news_encoder.skip_empty = False
        dense_empty = model.news_encoder((torch.ones_like(hx[:2]), torch.zeros_like(hist[1][:2])))
    assert (hm0 == 0).any() and (hm0 == 1).any()
    assert torch.equal(h0, h1) and torch.equal(hm0, hm1)
    assert torch.equal(r0, r1) and torch.equal(u0, u1) and torch.equal(c0, c1)
    assert torch.equal(all_empty[0], dense_empty[0]) and torch.equal(all_empty[1], dense_empty[1])


@pytest.mark.parametrize("tower", ["nrms", "additive_only"])
def test_unpadded_encoder_matches_padded(tower):
    """TextEncoder.unpadded (xnrs_text_encoder_fwd_unpadded): only the unmasked token rows are projected to
    queries / attended / out-projected / pooled; K and V for every row.  Prefix masks (the data's shape,
    dataset.py:77-85): bitwise equal to the padded kernels.  Masks with holes, fully masked and fully live news,
    the id-gather path and several passes: within the parity bar (only the pooling normaliser's summation order
    changes)."""
    from xnrs_amd import ops
    S, D, h, E = 50, 192, 4, 64
    att = layers.MultiHeadAttention(h, D) if tower == "nrms" else None
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 256), p_dropout=0.0, out_features=E,
                                             in_features=D, att=att), 131)
    rng = synth.rng_for(132)
    n = 300
    x = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32")).to(DEV)
    L = rng.integers(0, S + 1, size=n)
    L[:3] = (0, S, 1)
    m = torch.from_numpy((np.arange(S)[None, :] < L[:, None]).astype("float32")).to(DEV)
    with torch.no_grad():
        y0, hm0 = ops.text_encoder(x, m, enc)
        y1, hm1 = ops.text_encoder_unpadded(x, m, enc)
        assert torch.equal(hm0, hm1) and torch.equal(y0, y1)
        # several passes + the id-gather (table) path
        ids = torch.from_numpy(rng.integers(0, n, size=(421,)).astype("int32")).to(DEV)
        y2, hm2 = ops.text_encoder_forward_unpadded(x, m, enc.att, enc.pooler, enc.head, ids=ids, news_per_pass=100)
        assert torch.equal(y2, y0[ids.long()]) and torch.equal(hm2, hm0[ids.long()])
        # masks with holes
        mh = torch.from_numpy((rng.random((n, S)) < 0.5).astype("float32")).to(DEV)
        mh[0] = 0
        y3, hm3 = ops.text_encoder(x, mh, enc)
        y4, hm4 = ops.text_encoder_unpadded(x, mh, enc)
        assert torch.equal(hm3, hm4)
        H.assert_close(y4, y3, tol=1e-6, what="holey masks")
        # against the CPU oracle as well
        yo, _ = O.text_encoder(x[:40].cpu().unsqueeze(0), mh[:40].cpu().reshape(1, 40, S, 1), sd, h if att is not None else None)
        H.assert_close(y4[:40], yo.reshape(40, E), what="unpadded vs oracle")
        # the module switch, combined with skip_empty
        enc.unpadded = enc.skip_empty = True
        y5, hm5 = enc((x.reshape(6, 50, S, D), m.reshape(6, 50, S, 1)))
        y6, hm6 = enc.forward_ids(x, m, ids.reshape(1, -1).long(), dedup=True)  # table path through the same switch
        enc.unpadded = enc.skip_empty = False
        assert torch.equal(y5.reshape(n, E), y0) and torch.equal(hm5.reshape(n), hm0)
        assert torch.equal(y6[0], y0[ids.long()]) and torch.equal(hm6[0, :, 0], hm0[ids.long()])
        # a non-binary mask is refused, not mis-computed: the host-compacted path raises (it reads the counts anyway), the
        # device-compacted one -- no host read to raise from -- poisons its outputs with NaN
        from xnrs_amd.hip import XnrsHipError
        with pytest.raises(XnrsHipError):
            ops.text_encoder_forward_unpadded(x, m * 0.5, enc.att, enc.pooler, enc.head)
        if ops.compact_supported(S, D, enc.att, enc.pooler):  # (not under XNRS_FOLD_OUT=0 / the split GEMM modes)
            yb, hmb = ops.text_encoder_forward_compact(x, m * 0.5, enc.att, enc.pooler, enc.head)
            assert torch.isnan(yb).all() and torch.isnan(hmb).all()
            with pytest.raises(hip.XnrsHipError, match="mask value other than 0 / 1"):
                hip.check_status()  # ... and says so in the sticky status word (read and cleared here)


def test_naml_with_padding_free_encoders_matches_golden():
    """NAML (naml.py:61-147): title and abstract towers with skip_empty + unpadded switched on still reproduce the
    golden vectors of the real reference."""
    g = H.golden("models")
    name = next(k for k in sorted(cases.MODELS) if cases.MODELS[k]["model"] == "NAML")
    c = cases.MODELS[name]
    model, sd = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = cases.model_batch(c)
    for enc in (model.title_encoder, model.body_encoder):
        enc.skip_empty = enc.unpadded = True
    with torch.no_grad():
        r = model(batch)
        ue = model.get_user_embeddings(batch)
    H.assert_close(r, g[f"{name}/r"], what=name + " scores")
    H.assert_close(ue, g[f"{name}/ue"], what=name + " ue")


@pytest.mark.parametrize("S,D,h,A,bias,with_ids", [(50, 64, 4, 48, True, False), (40, 96, 3, 80, False, False),
                                                   (33, 128, 8, 256, True, True), (7, 20, 5, 16, True, False),
                                                   (64, 32, 2, 100, False, True)])
def test_folded_out_projection_matches_per_token_out_projection(S, D, h, A, bias, with_ids):
    """Inference folds the attention out-projection behind the pooling (api.hip "fold": fc1 on the O rows with W1.Wo, one
    Wo product per news after the weighted sum, bias times the sum of the weights) -- exact algebra for every input, a
    different rounding order.  Against the per-token out-projection (XNRS_FOLD_OUT=0, the reference's order) and the
    oracle: news vectors, news mask, with and without biases, the id-gather path, an all-masked and a fully live news,
    masks with holes; and the pooling WEIGHTS of the user tower (return_weights) under both orders."""
    from xnrs_amd import hip
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=24,
                                             in_features=D, att=layers.MultiHeadAttention(h, D), bias=bias), 501)
    rng = synth.rng_for(502 + S)
    n = 37
    x = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32"))
    m = torch.from_numpy((rng.random((n, S)) < 0.7).astype("float32"))
    m[0] = 0
    m[1] = 1
    ids = torch.from_numpy(rng.integers(0, n, size=(50,)).astype("int64")) if with_ids else None

    def run():
        with torch.no_grad():
            if with_ids:
                y, hm = enc.forward_ids(x.to(DEV), m.to(DEV), ids.to(DEV).reshape(1, -1))
            else:
                y, hm = enc((x.to(DEV).unsqueeze(0), m.to(DEV).reshape(1, n, S, 1)))
        return y[0], hm[0]

    with hip.knobs(XNRS_FOLD_OUT="2"):
        y1, hm1 = run()
    with hip.knobs(XNRS_FOLD_OUT="0"):
        y0, hm0 = run()
    assert torch.equal(hm0, hm1)
    H.assert_close(y1, y0, 2e-5, "folded vs per-token out-projection")
    assert not torch.equal(y1, y0) or D < 32  # (the two orders really are different computations)
    xo, mo = (x[ids], m[ids]) if with_ids else (x, m)
    yo, hmo = O.text_encoder(xo.unsqueeze(0), mo.reshape(1, -1, S, 1), sd, h)
    H.assert_close(y1, yo[0], what="folded vs oracle")
    assert torch.equal(hm1.cpu(), hmo[0])

    ue, usd = load(user_encoding.UserEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, emb_dim=D,
                                             att=layers.MultiHeadAttention(h, D), bias=bias), 503)
    hx = torch.from_numpy(rng.standard_normal((5, S, D)).astype("float32"))
    hmask = torch.from_numpy((rng.random((5, S, 1)) < 0.7).astype("float32"))
    with torch.no_grad():
        with hip.knobs(XNRS_FOLD_OUT="2"):
            u1, a1 = ue((hx.to(DEV), hmask.to(DEV)), return_weights=True)
        with hip.knobs(XNRS_FOLD_OUT="0"):
            u0, a0 = ue((hx.to(DEV), hmask.to(DEV)), return_weights=True)
    H.assert_close(u1, u0, 2e-5, "user vector, folded vs per-token")
    H.assert_close(a1, a0, 2e-5, "pooling weights, folded vs per-token")
    uo, ao = O.user_encoder(hx, hmask, usd, h, return_weights=True)
    H.assert_close(u1, uo, what="user vector vs oracle")
    H.assert_close(a1, ao, what="pooling weights vs oracle")


@pytest.mark.parametrize("S,D,h,A", [(50, 64, 4, 48), (40, 96, 3, 100), (33, 128, 8, 256), (9, 32, 2, 16), (64, 64, 2, 33)])
def test_fc2_dot_in_the_fc1_epilogue(S, D, h, A):
    """Inference takes the additive pooler's score w2 . tanh(fc1 x) per block of 32 hidden columns in the fc1 GEMM's
    epilogue (GemmArgs::rowdot_out: tanh(fc1 x) is never stored; the pooling kernel adds the blocks in column order).
    Against the materialised-T path (XNRS_FC1_ROWDOT=0) and the oracle: hidden sizes that are / are not multiples of 32,
    ragged row tiles, pooling weights of the user tower, and the padding-free path bitwise equal to the padded one."""
    from xnrs_amd import hip
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=24,
                                             in_features=D, att=layers.MultiHeadAttention(h, D)), 511)
    rng = synth.rng_for(512 + A)
    n = 41
    x = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32"))
    lens = rng.integers(0, S + 1, size=(n,))
    m = torch.from_numpy((np.arange(S)[None, :] < lens[:, None]).astype("float32"))  # prefix masks (the data's shape)

    def run(unpadded=False):
        enc.unpadded = unpadded
        try:
            with torch.no_grad():
                y, hm = enc((x.to(DEV).unsqueeze(0), m.to(DEV).reshape(1, n, S, 1)))
        finally:
            enc.unpadded = False
        return y[0], hm[0]

    with hip.knobs(XNRS_NEWS_FUSED="0"):  # (the GEMM pipeline, whatever the short-title dispatch would pick)
        y1, hm1 = run()
    with hip.knobs(XNRS_FC1_ROWDOT="0", XNRS_NEWS_FUSED="0"):
        y0, hm0 = run()
    assert torch.equal(hm0, hm1)
    H.assert_close(y1, y0, 2e-5, "epilogue dot vs materialised tanh(fc1 x)")
    yo, _ = O.text_encoder(x.unsqueeze(0), m.reshape(1, n, S, 1), sd, h)
    H.assert_close(y1, yo[0], what="epilogue dot vs oracle")
    if S <= 64 and (D // h) % 4 == 0:
        with hip.knobs(XNRS_NEWS_FUSED="0"):
            y2, hm2 = run(unpadded=True)
        assert torch.equal(y2, y1) and torch.equal(hm2, hm1)

    ue, usd = load(user_encoding.UserEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, emb_dim=D,
                                             att=layers.MultiHeadAttention(h, D)), 513)
    hx = torch.from_numpy(rng.standard_normal((6, S, D)).astype("float32"))
    hmask = torch.from_numpy((rng.random((6, S, 1)) < 0.7).astype("float32"))
    with torch.no_grad():
        u1, a1 = ue((hx.to(DEV), hmask.to(DEV)), return_weights=True)
    uo, ao = O.user_encoder(hx, hmask, usd, h, return_weights=True)
    H.assert_close(u1, uo, what="user vector vs oracle")
    H.assert_close(a1, ao, what="pooling weights vs oracle")


@pytest.mark.parametrize("tower", ["nrms", "additive_only"])
def test_device_compacted_encoder_and_graph_capture(tower):
    """xnrs_text_encoder_fwd_compact: the padding-free encoder with the live-row / K|V-row lists and their counts built
    on the DEVICE (the GEMMs read the row count from device memory).  Prefix masks: bitwise equal to the padded kernels
    and to the host-compacted path, over several passes, with a table + ids, with all-masked and fully live news; then
    the whole forward (skip_empty + unpadded switched on) is CAPTURED in a hipGraph -- no host sync anywhere -- and the
    replay on new data equals the dense step bit for bit."""
    from xnrs_amd import ops
    S, D, h, E = 50, 192, 4, 64
    att = layers.MultiHeadAttention(h, D) if tower == "nrms" else None
    enc, sd = load(news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 256), p_dropout=0.0, out_features=E,
                                             in_features=D, att=att), 231)
    if not (ops.COMPACT_ON_DEVICE and ops.compact_supported(S, D, att, enc.pooler)):
        pytest.skip("the device-compacted entry point is switched off by a development knob in this environment")
    rng = synth.rng_for(232)
    n = 700
    x = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32")).to(DEV)
    L = rng.integers(0, S + 1, size=n)
    L[:3] = (0, S, 1)
    L[100:160] = 0  # a run of empty news (history padding)
    m = torch.from_numpy((np.arange(S)[None, :] < L[:, None]).astype("float32")).to(DEV)
    with torch.no_grad():
        y0, hm0 = ops.text_encoder(x, m, enc)
        y1, hm1 = ops.text_encoder_forward_compact(x, m, enc.att, enc.pooler, enc.head)
        y2, hm2 = ops.text_encoder_forward_compact(x, m, enc.att, enc.pooler, enc.head, chunk=97)  # 8 ragged passes
        y3, hm3 = ops.text_encoder_forward_unpadded(x, m, enc.att, enc.pooler, enc.head)          # host-compacted
        assert torch.equal(hm1, hm0) and torch.equal(y1, y0)
        assert torch.equal(hm2, hm0) and torch.equal(y2, y0)
        assert torch.equal(hm3, hm0) and torch.equal(y3, y0)
        ids = torch.from_numpy(rng.integers(0, n, size=(1234,)).astype("int32")).to(DEV)
        y4, hm4 = ops.text_encoder_forward_compact(x, m, enc.att, enc.pooler, enc.head, ids=ids, chunk=300)
        assert torch.equal(y4, y0[ids.long()]) and torch.equal(hm4, hm0[ids.long()])
        mh = torch.from_numpy((rng.random((n, S)) < 0.5).astype("float32")).to(DEV)  # masks with holes
        mh[0] = 0
        y5, hm5 = ops.text_encoder(x, mh, enc)
        y6, hm6 = ops.text_encoder_forward_compact(x, mh, enc.att, enc.pooler, enc.head)
        assert torch.equal(hm5, hm6)
        H.assert_close(y6, y5, tol=1e-6, what="holey masks")
        # ---- capture: the module path with both padding-free switches on
        enc.unpadded = enc.skip_empty = True
        try:
            xs, ms = x.reshape(14, 50, S, D).clone(), m.reshape(14, 50, S, 1).clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                enc((xs, ms))  # warm-up outside the capture (workspace growth, fold cache)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yg, hmg = enc((xs, ms))
            # replay on DIFFERENT data (other tokens, other lengths): the launch sequence did not depend on the first batch
            L2 = rng.integers(0, S + 1, size=n)
            L2[5:40] = 0
            x2 = torch.from_numpy(rng.standard_normal((n, S, D)).astype("float32")).to(DEV)
            m2 = torch.from_numpy((np.arange(S)[None, :] < L2[:, None]).astype("float32")).to(DEV)
            xs.copy_(x2.reshape(14, 50, S, D))
            ms.copy_(m2.reshape(14, 50, S, 1))
            g.replay()
            torch.cuda.synchronize()
        finally:
            enc.unpadded = enc.skip_empty = False
        yd, hmd = ops.text_encoder(x2, m2, enc)
        assert torch.equal(yg.reshape(n, E), yd) and torch.equal(hmg.reshape(n), hmd)


def test_nonbinary_mask_on_the_device_compacted_path_sets_the_status_word():
    """The device-compacted encoder never synchronises, so a mask value other than 0 / 1 cannot raise from it: its outputs
    are NaN AND the sticky device status word (include/xnrs_hip.h: xnrs_set_status_word) carries XNRS_STATUS_NONBINARY_MASK
    until the caller reads it (hip.check_status()); ops.check_binary_mask is the set-up-time check that raises at once."""
    from xnrs_amd import hip, ops
    S, D, h, E = 20, 64, 4, 32
    enc = news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, 64), p_dropout=0.0, out_features=E, in_features=D,
                                    att=layers.MultiHeadAttention(h, D))
    enc = enc.eval().to(DEV)
    rng = synth.rng_for(9100)
    x, m = synth.token_block(rng, 1, 40, S, D, min_len=2)
    x, m = x.to(DEV), m.to(DEV)
    enc.unpadded = True
    hip.clear_status()  # (sticky by design: an earlier test may have left its bit)
    with torch.no_grad():
        y_ok, _ = enc((x, m))
        assert torch.isfinite(y_ok).all()
        hip.check_status()  # nothing to report
        bad = m.clone()
        bad[0, 3, 1, 0] = 0.5
        with pytest.raises(ValueError):
            ops.check_binary_mask(bad)
        ops.check_binary_mask(m)
        y_bad, _ = enc((x, bad))
        assert torch.isnan(y_bad).all()
    with pytest.raises(hip.XnrsHipError, match="mask value other than 0 / 1"):
        hip.check_status()
    hip.check_status()


@pytest.mark.parametrize("pool", ["additive", "mean"])
def test_attention_rows_of_masked_queries_are_not_computed_in_pooled_calls(pool):
    """Inside a pooled encoder call the attention core leaves all-masked sequences at once and writes zeros for query tiles
    whose 16 rows are all masked (round 4): both poolers multiply such rows by exactly 0 (layers.py:33,62-65), so news
    vectors and masks are BIT FOR BIT those of the call that computes every attention row (XNRS_MHA_SKIP_MASKED=0) -- and
    within the bar of the oracle.  Empty news, short titles (whole tiles masked), masks with holes, S = 50 / d_k = 16."""
    S, D, h, E = 50, 64, 4, 32
    pooler = layers.AdditiveAttention(D, 48) if pool == "additive" else layers.MaskedMean()
    enc = news_encoding.TextEncoder(pooler=pooler, p_dropout=0.0, out_features=E, in_features=D, att=layers.MultiHeadAttention(h, D))
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, 9201)
    enc.load_state_dict(sd)
    enc = enc.eval().to(DEV)
    rng = synth.rng_for(9200)
    n = 90
    x = torch.from_numpy(rng.standard_normal((1, n, S, D)).astype("float32"))
    lens = rng.integers(0, S + 1, size=n)
    lens[:6] = [0, 1, 15, 16, 17, 50]
    m = (torch.arange(S)[None, :] < torch.from_numpy(lens)[:, None]).to(torch.float32)
    m[40:50] *= torch.from_numpy((rng.random((10, S)) < 0.5).astype("float32"))  # holes
    m = m.reshape(1, n, S, 1)
    with torch.no_grad():
        y1, hm1 = enc((x.to(DEV), m.to(DEV)))
        with hip.knobs(XNRS_MHA_SKIP_MASKED="0"):
            y0, hm0 = enc((x.to(DEV), m.to(DEV)))
    assert torch.equal(y1, y0) and torch.equal(hm1, hm0)
    yo, hmo = O.text_encoder(x, m, sd, h)
    H.assert_close(y1, yo, what="news vectors vs oracle")


@pytest.mark.parametrize("name", ["NRMS", "standard"])
def test_one_news_encoder_call_for_a_small_request_is_bitwise_the_two_calls(name):
    """ParentRec._forward in inference: history and candidates of a small request go through ONE news-encoder call (a news
    vector depends on its own rows only) -- scores and embeddings bit for bit those of the reference's two calls; the same for
    the id path (one call over the ids of both sides)."""
    from tests.golden import cases
    from xnrs_amd import synth
    from xnrs_amd.models import make_model

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    c = dict(model=name, B=3, H=7, C=4, S=20, D=64, h=4, E=32, bias=True, seed=6101, min_len=3)
    model = make_model(Cfg(cases.model_cfg(c)))
    model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 6102))
    model = model.eval().to(DEV)
    batch = synth.batch_to(cases.model_batch(c), DEV)
    hist = batch["user_features"]["history"]["title_emb"]
    cand = batch["candidate_features"]["title_emb"]
    with torch.no_grad():
        assert model._one_news_call(hist, cand)
        r1, u1, c1 = model._forward(hist, cand, return_embeddings=True)
        model.ONE_CALL_MAX_BYTES = 0
        try:
            assert not model._one_news_call(hist, cand)
            r0, u0, c0 = model._forward(hist, cand, return_embeddings=True)
        finally:
            del model.ONE_CALL_MAX_BYTES
    assert torch.equal(r1, r0) and torch.equal(u1, u0) and torch.equal(c1, c0)
    assert not model.train()._one_news_call(hist, cand)  # (training keeps the reference's call structure)
    model.eval()
    # id path: a table of the batch's own news, ids = their positions
    B, H, C = c["B"], c["H"], c["C"]
    tx = torch.cat([hist[0].reshape(B * H, c["S"], c["D"]), cand[0].reshape(B * C, c["S"], c["D"])])
    tm = torch.cat([hist[1].reshape(B * H, c["S"]), cand[1].reshape(B * C, c["S"])])
    hid = torch.arange(B * H, device=DEV, dtype=torch.int32).reshape(B, H)
    cid = (B * H + torch.arange(B * C, device=DEV, dtype=torch.int32)).reshape(B, C)
    with torch.no_grad():
        ri = model.forward_ids(tx, tm, hid, cid)
    assert torch.equal(ri, r0)
