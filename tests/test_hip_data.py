"""GPU: device-side batch assembly, CSR scoring, ranking metrics and the evaluation path against the data
oracle / the model oracle."""
import numpy as np
import pytest
import torch

from oracle import data_oracle as DO
from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases
from xnrs_amd import evaluation as EV
from xnrs_amd import synth
from xnrs_amd.data import Behaviors, DeviceBatcher, NewsStore
from xnrs_amd.models import make_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg(dict):
    __getattr__ = dict.__getitem__


def setup():
    news_feat, sessions = cases.data_corpus()
    store = NewsStore.from_news_feat(news_feat, "title_emb", ["category_index"])
    beh = Behaviors.from_sessions(sessions, store)
    return news_feat, sessions, store, beh


def model_for(c):
    mc = dict(model="NRMS", B=1, H=c["l_hist"], C=1, S=c["S"], D=c["D"], h=2, E=8, bias=False)
    model = make_model(Cfg(cases.model_cfg(mc)))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 611)
    model.load_state_dict(sd)
    return model.eval().to(DEV), sd, mc


def test_train_batch_bit_exact_vs_oracle():
    news_feat, sessions, store, beh = setup()
    c = cases.DATA
    bat = DeviceBatcher(beh.to(DEV), c["l_hist"])
    sess = torch.tensor([5, 0, 3, 3, 1, 4, 2], device=DEV)
    hist, cand, targets = bat.train_batch(sess, c["n_neg"], seed=77)
    for b, si in enumerate(sess.tolist()):
        s = sessions[si]
        h, cd = DO.train_rows(si, store.rows(s["history"]), store.rows(s["positives"]), store.rows(s["negatives"]),
                              c["l_hist"], c["n_neg"], seed=77)
        assert hist[b].tolist() == h and cand[b].tolist() == cd
    assert targets[:, 0].min() == 1 and targets[:, 1:].max() == 0
    h2, c2, _ = bat.train_batch(sess, c["n_neg"], seed=78)
    assert torch.equal(h2, hist) and not torch.equal(c2, cand)


def test_eval_batch_and_forward_ids_equal_dense_forward():
    news_feat, sessions, store, beh = setup()
    c = cases.DATA
    model, sd, mc = model_for(c)
    dstore = store.to(DEV)
    bat = DeviceBatcher(beh.to(DEV), c["l_hist"])
    sess = torch.arange(len(sessions), device=DEV)
    hist, off, rows, csess, targets = bat.eval_batch(sess)
    x, m = store.x.numpy(), store.m.numpy()
    for b, s in enumerate(sessions):
        h, cd, t = DO.eval_rows(store.rows(s["history"]), store.rows(s["positives"]), store.rows(s["negatives"]), c["l_hist"])
        lo, hi = int(off[b]), int(off[b + 1])
        assert hist[b].tolist() == h and rows[lo:hi].tolist() == cd and targets[lo:hi].tolist() == t
        assert csess[lo:hi].unique().tolist() == [b]
        # id path == dense path (what the reference's dataset would have handed over), bit for bit
        with torch.no_grad():
            r_ids = model.forward_ids(dstore.x, dstore.m, hist[b:b + 1], rows[lo:hi].reshape(1, -1))
            hx, hm = DO.materialise(x, m, h)
            cx, cm = DO.materialise(x, m, cd)
            r_dense = model._forward((torch.from_numpy(hx)[None], torch.from_numpy(hm)[None]),
                                     (torch.from_numpy(cx)[None], torch.from_numpy(cm)[None]))
        assert torch.equal(r_ids, r_dense)
        ro = O.parent_forward((torch.from_numpy(hx)[None], torch.from_numpy(hm)[None]),
                              (torch.from_numpy(cx)[None], torch.from_numpy(cm)[None]), sd, mc["h"])
        H.assert_close(r_ids, ro)


def test_rank_metrics_vs_oracle():
    rng = np.random.default_rng(5)
    ts, ss, off = [], [], [0]
    for name, (t, s) in cases.METRIC_CASES.items():
        ts += t
        ss += s
        off.append(len(ts))
    for _ in range(40):  # random impressions, relu-style ties included
        C = int(rng.integers(2, 90))
        t = np.zeros(C)
        t[rng.choice(C, size=int(rng.integers(1, max(2, C // 4))), replace=False)] = 1
        s = np.maximum(rng.standard_normal(C), 0.0) if rng.random() < 0.5 else rng.random(C)
        if t.min() == 1:
            t[0] = 0
        ts += t.tolist()
        ss += s.tolist()
        off.append(len(ts))
    out = EV.rank_metrics(torch.tensor(ss, dtype=torch.float32, device=DEV), torch.tensor(ts, dtype=torch.float32, device=DEV),
                          torch.tensor(off, dtype=torch.int64, device=DEV)).cpu().numpy()
    g = H.golden("data")
    for b in range(len(off) - 1):
        t = np.array(ts[off[b]:off[b + 1]])
        s = np.array(ss[off[b]:off[b + 1]], dtype=np.float32).astype(np.float64)
        ref = DO.impression_metrics(t, s)
        assert np.allclose(out[b], ref, rtol=2e-6, atol=2e-6), (b, out[b], ref)
    for b, name in enumerate(cases.METRIC_CASES):
        if len(cases.METRIC_CASES[name][0]) <= 16 and name != "relu_ties":  # tie-free: must equal the REAL reference
            assert np.allclose(out[b], g[f"metrics/{name}"], rtol=2e-6, atol=2e-6), name


def test_evaluate_end_to_end():
    """evaluate(): encode every news once, score CSR, metrics on device == the reference-style loop
    (re-encode per impression with the CPU oracle, metrics with the data oracle)."""
    news_feat, sessions, store, beh = setup()
    c = cases.DATA
    model, sd, mc = model_for(c)
    res = EV.evaluate(model, store.to(DEV), beh.to(DEV), c["l_hist"], batch=4)
    x, m = store.x.numpy(), store.m.numpy()
    acc = np.zeros(9)
    for s in sessions:
        h, cd, t = DO.eval_rows(store.rows(s["history"]), store.rows(s["positives"]), store.rows(s["negatives"]), c["l_hist"])
        hx, hm = DO.materialise(x, m, h)
        cx, cm = DO.materialise(x, m, cd)
        r = torch.relu(O.parent_forward((torch.from_numpy(hx)[None], torch.from_numpy(hm)[None]),
                                        (torch.from_numpy(cx)[None], torch.from_numpy(cm)[None]), sd, mc["h"]))
        acc += DO.impression_metrics(t, r.reshape(-1).numpy())
    acc /= len(sessions)
    got = np.array([res[k] for k in EV.METRIC_NAMES])
    assert np.allclose(got, acc, rtol=1e-4, atol=1e-5), (got, acc)


def test_host_tensors_fail_loudly():
    from xnrs_amd.hip import XnrsHipError
    news_feat, sessions, store, beh = setup()
    with pytest.raises(XnrsHipError):
        DeviceBatcher(beh, 5)


def test_dedup_equals_plain_id_path():
    news_feat, sessions, store, beh = setup()
    c = cases.DATA
    model, sd, mc = model_for(c)
    dstore = store.to(DEV)
    bat = DeviceBatcher(beh.to(DEV), c["l_hist"])
    sess = torch.tensor([0, 1, 2, 3, 4, 5, 1, 1], device=DEV)
    hist, cand, _ = bat.train_batch(sess, c["n_neg"], seed=3)
    with torch.no_grad():
        r0, u0, c0 = model.forward_ids(dstore.x, dstore.m, hist, cand, return_embeddings=True)
        r1, u1, c1 = model.forward_ids(dstore.x, dstore.m, hist, cand, return_embeddings=True, dedup=True)
    assert torch.equal(r0, r1) and torch.equal(u0, u1) and torch.equal(c0, c1)
    # gradients flow through the dedup scatter as well
    model.train()
    r = model.forward_ids(dstore.x, dstore.m, hist, cand, dedup=True)
    r.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters() if "dummy" not in n)


def test_gather_rows_equals_the_dataset_materialisation():
    """NewsStore.gather (xnrs_gather_rows) = what NewsRecDataset.__getitem__ would have concatenated on the host
    (oracle: data_oracle.materialise, pinned to the real dataset by tests/golden/data.npz), bit for bit -- vector and
    scalar code paths (S*D % 4 == 0 with S % 4 != 0 for the mask), repeated and empty-slot rows, 2-d row arrays."""
    news_feat, sessions, store, beh = setup()
    dstore = store.to(DEV)
    rng = synth.rng_for(911)
    rows = torch.from_numpy(rng.integers(0, len(store.ids) + 1, size=(3, 7)).astype(np.int32))
    rows[0, 0] = 0  # the empty slot
    x, m = dstore.gather(rows.to(DEV))
    hx, hm = DO.materialise(store.x.numpy(), store.m.numpy(), rows.reshape(-1).tolist())
    assert x.shape == (3, 7) + tuple(store.x.shape[1:]) and m.shape == (3, 7, store.x.shape[1], 1)
    assert np.array_equal(x.cpu().numpy().reshape(hx.shape), hx) and np.array_equal(m.cpu().numpy().reshape(hm.shape), hm)
    # a big block shape (150 KB rows, the shipped token shape) against torch indexing
    gen = torch.Generator(device=DEV)
    gen.manual_seed(1)
    tx, tm = synth.device_tokens(gen, 64, 50, 768, DEV)
    big = NewsStore(tx, tm.reshape(64, 50), list(range(63)))
    ids = torch.randint(0, 64, (200,), generator=gen, device=DEV, dtype=torch.int32)
    gx, gm = big.gather(ids)
    assert torch.equal(gx, tx[ids.long()]) and torch.equal(gm, tm[ids.long()])
    with pytest.raises(Exception):
        store.gather(rows)  # host tensors fail loudly
