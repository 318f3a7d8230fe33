"""GPU: BASELINE configs[3]/[4] as first-class citizens of the device-resident data path.

* NAML and the LSTUR news tower by TABLE ROW (two token tables + two category columns gathered by the same news id,
  dataset.py:63-124) against the goldens the REAL NewsRecDataset + REAL NAML / LSTURNewsEncoder produced
  (tests/golden/naml_ids.npz), bitwise against the dense path on the materialised batch, with dedup, through the
  evaluation epoch, and starting from the on-disk store (file -> HBM loader);
* full-size property tests for NAML and StandardRec at B=512, H=25, C=5, S=50, D=768."""
import numpy as np
import pytest
import torch

from oracle import data_oracle as DO
from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases
from tests.test_naml_ids_oracle import corpus_store, naml_state, oracle_batch
from xnrs_amd import evaluation as EV
from xnrs_amd import synth
from xnrs_amd.data import Behaviors, DeviceBatcher, NewsStore
from xnrs_amd.models import make_model
from xnrs_amd.models.full_models import LSTURNewsEncoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg(dict):
    __getattr__ = dict.__getitem__


def naml_model():
    c = cases.NAML_DATA
    model = make_model(Cfg(cases.naml_data_cfg(c)))
    sd = naml_state(c)
    model.load_state_dict(sd)
    return model.eval().to(DEV), sd


def lstur_model():
    c = cases.NAML_DATA
    cfg = Cfg(dict(cases.naml_data_cfg(c), catg_features=["category_index", "subcategory_index"]))
    enc = LSTURNewsEncoder(cfg)
    sd = naml_state(c, 1)
    enc.load_state_dict(sd)
    return enc.eval().to(DEV), sd


def check_sessions(model, dstore, store, sessions, lstur=None):
    """Every eval session: id path == golden (real dataset + real model), == dense path on the materialised batch bit
    for bit, dedup == plain bit for bit."""
    c = cases.NAML_DATA
    g = H.golden("naml_ids")
    beh = Behaviors.from_sessions(sessions, store).to(DEV)
    bat = DeviceBatcher(beh, c["l_hist"])
    hist, off, rows, csess, targets = bat.eval_batch(torch.arange(len(sessions), device=DEV))
    for i, s in enumerate(sessions):
        lo, hi = int(off[i]), int(off[i + 1])
        hid, cid = hist[i:i + 1], rows[lo:hi].reshape(1, -1)
        batch, h, cd = oracle_batch(store, s, c["l_hist"])
        assert hid[0].tolist() == h and cid[0].tolist() == cd
        with torch.no_grad():
            r = model.forward_store(dstore, hid, cid)
            r_dd, u_dd, _ = model.forward_store(dstore, hid, cid, dedup=True, return_embeddings=True)
            r_dense = model(synth.batch_to(batch, DEV))
            ue_dense = model.get_user_embeddings(synth.batch_to(batch, DEV))
        H.assert_close(r, g[f"naml_ids/s{i}/r"], what=f"s{i} scores vs the reference")
        H.assert_close(u_dd, g[f"naml_ids/s{i}/ue"], what=f"s{i} user vs the reference")
        assert torch.equal(r, r_dense), f"s{i}: id path != dense path"
        assert torch.equal(r_dd, r) and torch.equal(u_dd, ue_dense), f"s{i}: dedup != plain"
        if lstur is not None:
            with torch.no_grad():
                e, m = lstur.forward_ids(dstore, hid)
                hb = synth.batch_to(batch["user_features"]["history"], DEV)
                e_dense, m_dense = lstur(hb["title_emb"], hb["category_index"], hb["subcategory_index"])
            H.assert_close(e, g[f"lstur_ids/s{i}/e"], what=f"s{i} lstur news vectors vs the reference")
            assert np.array_equal(m.cpu().numpy(), g[f"lstur_ids/s{i}/m"])
            assert torch.equal(e, e_dense) and torch.equal(m, m_dense)


def test_naml_and_lstur_news_by_table_row():
    _, sessions, store = corpus_store()
    model, _ = naml_model()
    lstur, _ = lstur_model()
    check_sessions(model, store.to(DEV), store, sessions, lstur)


def test_naml_from_the_on_disk_store(tmp_path):
    """file -> HBM (NewsStore.load_to_device: memory map + two pinned staging buffers, chunks smaller than the table)
    -> gather == the file's bytes; file -> forward_store == the golden / dense path."""
    _, sessions, store = corpus_store()
    p = str(tmp_path / "corpus")
    store.save(p)
    stats = {}
    dstore = NewsStore.load_to_device(p, DEV, rows_per_chunk=7, stats=stats)
    assert stats["bytes"] == sum(t.numel() * 4 + m.numel() for t, m in (store.text("title_emb"), store.text("abstract_emb")))
    assert dstore.x.is_cuda and dstore.column("category_index").is_cuda and dstore.ids == [str(i) for i in store.ids]
    for feat in ("title_emb", "abstract_emb"):
        fx = np.fromfile(p + (".x.f32" if feat == "title_emb" else f".{feat}.x.f32"), dtype=np.float32).reshape(store.text(feat)[0].shape)
        fm = np.fromfile(p + (".m.u8" if feat == "title_emb" else f".{feat}.m.u8"), dtype=np.uint8).reshape(store.text(feat)[1].shape)
        rows = torch.tensor([[3, 0, 24], [7, 7, 1]], dtype=torch.int32, device=DEV)
        x, m = dstore.gather(rows, feat)
        assert np.array_equal(x.cpu().numpy(), fx[rows.cpu().numpy()])
        assert np.array_equal(m.cpu().numpy()[..., 0], fm[rows.cpu().numpy()].astype(np.float32))
    with pytest.raises(IndexError):  # the blocking check (two host reads) on request ...
        dstore.gather(torch.tensor([25], dtype=torch.int32, device=DEV), trusted=False)
    # ... by default no host read: the id is clamped (no out-of-bounds access) and the sticky status word says so later
    from xnrs_amd import hip
    hip.clear_status()  # (sticky by design: an earlier test may have left its bit)
    dstore.gather(torch.tensor([3, 25, -2], dtype=torch.int32, device=DEV))
    with pytest.raises(hip.XnrsHipError, match="row id outside the table"):
        hip.check_status()
    hip.check_status()  # read and cleared
    model, _ = naml_model()
    check_sessions(model, dstore, store, sessions)


def test_naml_evaluation_epoch():
    """evaluate() with the per-model hooks: every NAML news (both token tables + both category columns) encoded ONCE,
    users from the pre-encoded vectors, CSR scoring, metrics on the device == the reference-style loop (oracle model on
    the materialised batch per impression, relu, data-oracle metrics)."""
    c = cases.NAML_DATA
    _, sessions, store = corpus_store()
    model, sd = naml_model()
    beh = Behaviors.from_sessions(sessions, store)
    res = EV.evaluate(model, store.to(DEV), beh.to(DEV), c["l_hist"], batch=2)
    vecs, hm = EV.encode_news_table(model, store.to(DEV), rows_per_call=6)
    vecs1, hm1 = EV.encode_news_table(model, store.to(DEV))
    assert torch.equal(vecs, vecs1) and torch.equal(hm, hm1) and vecs.shape == (store.n_rows, c["E"])
    acc = np.zeros(len(EV.METRIC_NAMES))
    for s in sessions:
        batch, h, cd = oracle_batch(store, s, c["l_hist"])
        with torch.no_grad():
            r = torch.relu(O.naml_forward(batch, sd)).reshape(-1).numpy()
        t = np.array([1.0] * len(s["positives"]) + [0.0] * len(s["negatives"]))
        acc += DO.impression_metrics(t, r)
    for k, v in zip(EV.METRIC_NAMES, acc / len(sessions)):
        assert abs(res[k] - v) <= 1e-5, (k, res[k], v)


def big_model(name, seed=99):
    c = dict(model=name, E=256, bias=False, h=16, D=768, H=25, S=50)
    model = make_model(Cfg(synth.model_cfg(c)))
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd)
    return model.eval().to(DEV), sd


def big_batch(name, B, H_, C, S=50, D=768, seed=11):
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)

    def toks(n, ragged):
        x, m = synth.device_tokens(gen, B * n, S, D, DEV)
        if ragged:  # trailing history slots empty (all-zero x and m, dataset.py:82-85)
            n_hist = torch.randint(1, n + 1, (B, 1), generator=gen, device=DEV)
            live = (torch.arange(n, device=DEV)[None, :] < n_hist).reshape(B * n, 1, 1).float()
            x, m = x * live, m * live
        return x.reshape(B, n, S, D), m.reshape(B, n, S, 1)
    hist, cand = {"title_emb": toks(H_, True)}, {"title_emb": toks(C, False)}
    if name == "NAML":
        hist["abstract_emb"], cand["abstract_emb"] = toks(H_, True), toks(C, False)
        for d_, n in ((hist, H_), (cand, C)):
            d_["category_index"] = torch.randint(1, 20, (B, n), generator=gen, device=DEV, dtype=torch.int32)
            d_["subcategory_index"] = torch.randint(1, 301, (B, n), generator=gen, device=DEV, dtype=torch.int32)
    return {"user_features": {"history": hist, "other": {}}, "candidate_features": cand}


def cut(v, fn):
    if isinstance(v, torch.Tensor):
        return fn(v)
    if isinstance(v, dict):
        return {k: cut(x, fn) for k, x in v.items()}
    if isinstance(v, tuple):
        return tuple(cut(x, fn) for x in v)
    return v


@pytest.mark.parametrize("name", ["standard", "NAML"])
def test_full_size_properties_additive_models(name):
    """BASELINE configs[3] (StandardRec, the CL bi-encoder) and configs[4] (NAML) at B=512, H=25, C=5, S=50, D=768:
    * impressions are independent: the full step equals, bit for bit, the same impressions in ragged sub-batches;
    * permuting the candidates of an impression permutes its scores;
    * a slice of the full-size result matches the CPU oracle within the parity bar."""
    B, H_, C = 512, 25, 5
    model, sd = big_model(name)
    batch = big_batch(name, B, H_, C)
    with torch.no_grad():
        full = model(batch)
        assert full.shape == (B, C, 1) and torch.isfinite(full).all()
        parts, b0 = [], 0
        for nb in (37, 200, 1, 274):
            lo = b0
            parts.append(model(cut(batch, lambda t: t[lo:lo + nb])))
            b0 += nb
        assert b0 == B and torch.equal(torch.cat(parts), full)
        perm = torch.tensor([3, 0, 4, 1, 2], device=DEV)
        pb = dict(batch, candidate_features=cut(batch["candidate_features"], lambda t: t[:, perm]))
        assert torch.equal(model(pb), full[:, perm])
        sl = slice(100, 104)
        small = cut(batch, lambda t: t[sl].cpu())
        if name == "NAML":
            ref = O.naml_forward(small, sd)
        else:
            ref = O.parent_forward(small["user_features"]["history"]["title_emb"], small["candidate_features"]["title_emb"], sd, 16)
    H.assert_close(full[sl], ref, what=f"{name} full-size slice vs oracle")


def test_naml_table_scale_ids():
    """configs[4] the way DESIGN.md section 3 says large tables run: a device-resident corpus (4 096 news x title + abstract
    x 50 x 768 + both category columns = 2.5 GB), B=512 impressions as row ids, Zipf-distributed.  id path == dense path
    on the gathered batch bit for bit; dedup == plain bit for bit; a slice vs the CPU oracle."""
    n_news, B, H_, C, S, D = 4096, 512, 25, 5, 50, 768
    model, sd = big_model("NAML")
    gen = torch.Generator(device=DEV)
    gen.manual_seed(21)
    tx, tm = synth.device_tokens(gen, n_news + 1, S, D, DEV)
    ax, am = synth.device_tokens(gen, n_news + 1, S, D, DEV)
    for t in (tx, tm, ax, am):
        t[0] = 0
    cols = {"category_index": torch.randint(1, 20, (n_news + 1,), generator=gen, device=DEV, dtype=torch.int32),
            "subcategory_index": torch.randint(1, 301, (n_news + 1,), generator=gen, device=DEV, dtype=torch.int32)}
    cols["category_index"][0] = 0
    cols["subcategory_index"][0] = 0
    store = NewsStore(tx, tm.reshape(n_news + 1, S), list(range(n_news)), cols, {"abstract_emb": (ax, am.reshape(n_news + 1, S))})
    rng = np.random.default_rng(5)
    z = np.minimum(rng.zipf(1.1, size=(B, H_ + C)), n_news).astype(np.int32)
    n_hist = rng.integers(1, H_ + 1, size=(B, 1))
    z[:, :H_][np.arange(H_)[None, :] >= n_hist] = 0
    ids = torch.from_numpy(z).to(DEV)
    hid, cid = ids[:, :H_].contiguous(), ids[:, H_:].contiguous()
    with torch.no_grad():
        r = model.forward_store(store, hid, cid)
        r_dd = model.forward_store(store, hid, cid, dedup=True)

        def side(rows):
            out = {"title_emb": store.gather(rows, "title_emb"), "abstract_emb": store.gather(rows, "abstract_emb")}
            out["category_index"] = store.gather_column("category_index", rows)
            out["subcategory_index"] = store.gather_column("subcategory_index", rows)
            return out
        sl = slice(200, 264)
        dense = {"user_features": {"history": side(hid[sl]), "other": {}}, "candidate_features": side(cid[sl])}
        r_dense = model(dense)
        ref = O.naml_forward(cut(dense, lambda t: t[:3].cpu()), sd)
    assert torch.isfinite(r).all() and torch.equal(r_dd, r)
    assert torch.equal(r[sl], r_dense)
    H.assert_close(r[sl][:3], ref, what="NAML table-scale slice vs oracle")
