#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REAL reference implementation.

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.
The reference's hot-path modules are imported with the recipe of SURVEY.md section 8c (the two package
``__init__`` files that pull in dotmap/wandb are skipped by pre-seeding ``sys.modules``).

What is stored: ONLY outputs (numpy arrays) + the generator parameters.  Inputs and weights are
regenerated from PCG64 seeds by ``xnrs_amd.synth`` on every machine, so the fixtures stay small
and contain no reference source.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

for name, path in [("xnrs", f"{REF}/xnrs"), ("xnrs.models", f"{REF}/xnrs/models")]:
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    sys.modules[name] = mod

from xnrs.models.components import layers, news_encoding, user_encoding, scoring  # noqa: E402
from xnrs.models.full_models import NRMS, NAML, StandardRec, BaseRec  # noqa: E402
from xnrs.models.full_models.lstur import LSTURNewsEncoder  # noqa: E402
from xnrs.models.make_model import make_model  # noqa: E402

# xnrs/training.py imports two absent third-party packages that are not on the arithmetic path
# (SURVEY.md section 8c recipe 2): give them empty stand-ins so the reference's own InfoNCE
# (ContrastiveRankingTrainer._compute_contrastive_loss, training.py:433-472) can be called.
import importlib.machinery  # noqa: E402

for name, attrs in [("omegaconf", {"DictConfig": dict}), ("wandb", {})]:
    if name not in sys.modules:
        mod = types.ModuleType(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
from xnrs.training import ContrastiveRankingTrainer  # noqa: E402

from xnrs_amd import synth  # noqa: E402
from tests.golden import cases  # noqa: E402


def reference_infonce(emb, labels, temperature):
    """Call the reference's own loss with a bare ``self`` that only carries the temperature."""
    holder = types.SimpleNamespace(temperature=temperature)
    return ContrastiveRankingTrainer._compute_contrastive_loss(holder, emb, labels)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def load(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed)
    module.load_state_dict(sd)
    module.eval()
    return module


def npy(t):
    return t.detach().cpu().numpy()


def block_cases():
    out = {}
    for name, c in cases.BLOCKS.items():
        D = c["D"]
        x, m, u = cases.block_inputs(c)
        kind = c["kind"]
        with torch.no_grad():
            if kind == "additive":
                mod = load(layers.AdditiveAttention(D, c["A"]), c["seed"] + 1)
                y, a = mod(x, m if c["mask"] else None, return_weights=True)
                out[f"{name}/y"], out[f"{name}/a"] = npy(y), npy(a)
            elif kind == "mha":
                mod = load(layers.MultiHeadAttention(c["h"], D), c["seed"] + 1)
                out[f"{name}/y"] = npy(mod(x, m if c["mask"] else None))
            elif kind == "mean":
                out[f"{name}/y"] = npy(layers.MaskedMean()(x, m))
            elif kind == "dot":
                out[f"{name}/y"] = npy(scoring.DotScoring(normalize=c["normalize"])(u, x))
            else:
                raise ValueError(kind)
    return out


def encoder_cases():
    out = {}
    for name, c in cases.ENCODERS.items():
        D, E, A = c["D"], c["E"], c["A"]
        x, m = cases.encoder_inputs(c)
        att = layers.MultiHeadAttention(c["h"], D) if c["att"] else None
        pooler = layers.AdditiveAttention(D, A) if c["pooler"] == "additive" else layers.MaskedMean()
        with torch.no_grad():
            if c["tower"] == "news":
                enc = load(news_encoding.TextEncoder(pooler=pooler, p_dropout=0.0, out_features=E, in_features=D,
                                                     head=c["head"], att=att, bias=c["bias"]), c["seed"] + 1)
                y, hm = enc((x, m))
                out[f"{name}/y"], out[f"{name}/hm"] = npy(y), npy(hm)
            else:
                enc = load(user_encoding.UserEncoder(pooler=pooler, p_dropout=0.0, emb_dim=D, att=att,
                                                     head=c["head"], bias=c["bias"]), c["seed"] + 1)
                if c["pooler"] == "additive":
                    y, a = enc((x, m), None, return_weights=True)
                    out[f"{name}/a"] = npy(a)
                else:
                    y = enc((x, m))
                out[f"{name}/y"] = npy(y)
    return out


def model_cases():
    out = {}
    for name, c in cases.MODELS.items():
        cfg = Cfg(cases.model_cfg(c))
        model = load(make_model(cfg), c["seed"] + 1)
        batch = cases.model_batch(c)
        with torch.no_grad():
            if c["model"] == "NAML":
                out[f"{name}/r"] = npy(model(batch))
                out[f"{name}/ue"] = npy(model.get_user_embeddings(batch))
            else:
                r, u, cc = model(batch, return_embeddings=True)
                out[f"{name}/r"], out[f"{name}/u"], out[f"{name}/c"] = npy(r), npy(u), npy(cc)
                out[f"{name}/ue"] = npy(model.get_user_embeddings(batch))
    return out


def lstur_case():
    c = cases.LSTUR
    cfg = Cfg(cases.model_cfg(c))
    cfg["catg_features"] = ["category_index", "subcategory_index"]
    enc = load(LSTURNewsEncoder(cfg), c["seed"] + 1)
    x, m, ci, si = cases.lstur_inputs(c)
    with torch.no_grad():
        e, mm = enc((x, m), ci, si)
    return {"lstur_news/e": npy(e), "lstur_news/m": npy(mm)}


def grad_cases():
    """Loss value + gradients of the reference's train-step loss (training.py:402-472) in eval-mode
    dropout, computed with the reference's own modules and the reference's own per-row InfoNCE loop."""
    out = {}
    c = cases.GRAD
    cfg = Cfg(cases.model_cfg(c))
    model = load(make_model(cfg), c["seed"] + 1)
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx.requires_grad_(True)
    cx.requires_grad_(True)
    labels = cases.theme_labels(batch["main_theme"])
    preds = torch.relu(model(batch))
    loss_rec = torch.nn.functional.mse_loss(preds, batch["targets"])
    ue = model.get_user_embeddings(batch)
    loss_cl = reference_infonce(ue, labels, c["temperature"])
    loss = loss_rec + c["lambda_cl"] * loss_cl
    loss.backward()
    out["grad/loss"] = npy(loss)
    out["grad/loss_rec"] = npy(loss_rec)
    out["grad/loss_cl"] = npy(loss_cl)
    out["grad/d_hist_x"] = npy(hx.grad)
    out["grad/d_cand_x"] = npy(cx.grad)
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[f"grad/dW/{k}"] = npy(p.grad)
    # stand-alone InfoNCE value + gradient (B=8, E=16)
    e, lab = cases.infonce_inputs()
    e.requires_grad_(True)
    l = reference_infonce(e, lab, 0.08)
    l.backward()
    out["infonce/loss"], out["infonce/grad"] = npy(l), npy(e.grad)
    return out


def grad_shipped_cases(c=None, pre="gs"):
    """The train step of grad_cases() at the shipped NRMS shape (cases.GRAD_SHIPPED): loss terms whole, every gradient
    whole or as its fixed sample (cases.grad_sample).  Two of the three impressions share a theme, so the InfoNCE term is live
    (with two impressions its only negative IS the positive: loss 0)."""
    out = {}
    c = c or cases.GRAD_SHIPPED
    model = load(make_model(Cfg(cases.model_cfg(c))), c["seed"] + 1)
    batch = cases.model_batch(c)
    hx, hm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    hx.requires_grad_(True)
    cx.requires_grad_(True)
    labels = cases.theme_labels(c["themes"])
    preds = torch.relu(model(batch))
    loss_rec = torch.nn.functional.mse_loss(preds, batch["targets"])
    loss_cl = reference_infonce(model.get_user_embeddings(batch), labels, c["temperature"])
    loss = loss_rec + c["lambda_cl"] * loss_cl
    loss.backward()
    out[f"{pre}/loss"], out[f"{pre}/loss_rec"], out[f"{pre}/loss_cl"] = npy(loss), npy(loss_rec), npy(loss_cl)
    out[f"{pre}/d_hist_x"], out[f"{pre}/d_cand_x"] = cases.grad_sample(hx.grad), cases.grad_sample(cx.grad)
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[f"{pre}/dW/{k}"] = cases.grad_sample(p.grad)
            out[f"{pre}/max/{k}"] = npy(p.grad.abs().max())  # scale of the whole tensor (a sample may miss its largest entries)
    return out


def data_cases():
    """Outputs of the REAL NewsRecDataset / custom_collate_fn / evaluation.metrics on the tiny corpus."""
    import random
    from xnrs.data.dataset import NewsRecDataset
    from xnrs.utils import custom_collate_fn
    from xnrs.evaluation import metrics as M
    c = cases.DATA
    news_feat, sessions = cases.data_corpus(c)
    out = {}
    kw = dict(l_seq=c["S"], l_hist=c["l_hist"], text_features=["title_emb"], catg_features=["category_index"])
    ev = NewsRecDataset(uds=sessions, news_feat=news_feat, mode="eval", n_negatives=None, **kw)
    for i in range(len(sessions)):
        it = ev[i]
        hx, hm = it["user_features"]["history"]["title_emb"]
        cx, cm = it["candidate_features"]["title_emb"]
        out[f"data/eval{i}/hx"], out[f"data/eval{i}/hm"] = npy(hx), npy(hm)
        out[f"data/eval{i}/cx"], out[f"data/eval{i}/cm"] = npy(cx), npy(cm)
        out[f"data/eval{i}/t"] = npy(it["targets"])
        out[f"data/eval{i}/hcat"] = npy(it["user_features"]["history"]["category_index"])
        out[f"data/eval{i}/ccat"] = npy(it["candidate_features"]["category_index"])
    tr = NewsRecDataset(uds=sessions, news_feat=news_feat, mode="train", n_negatives=c["n_neg"], **kw)
    random.seed(1234)
    items = [tr[i] for i in range(len(sessions))]
    chosen = [it["item_ids"] for it in items]
    batch = custom_collate_fn(items)
    bx, bm = batch["user_features"]["history"]["title_emb"]
    cx, cm = batch["candidate_features"]["title_emb"]
    out["data/train/hx"], out["data/train/hm"] = npy(bx), npy(bm)
    out["data/train/cx"], out["data/train/cm"] = npy(cx), npy(cm)
    out["data/train/t"] = npy(batch["targets"])
    ids = list(news_feat)
    out["data/train/chosen"] = np.array([[ids.index(n) for n in ch] for ch in chosen], dtype=np.int32)
    for name, (t, s_) in cases.METRIC_CASES.items():
        t, s_ = np.array(t, dtype=np.float64), np.array(s_, dtype=np.float64)
        out[f"metrics/{name}"] = np.array([M.ndcg_score(t, s_, k=5), M.ndcg_score(t, s_, k=10), M.rr_score(t, s_),
                                           M.ctr_score(t, s_, k=1), M.ctr_score(t, s_, k=10), M.auc_score(t, s_),
                                           M.acc_score(t, s_), M.recall_score(t, s_), M.precision_score(t, s_)])
    return out


def naml_data_cases():
    """The REAL NewsRecDataset (eval mode: history truncation / zero padding, all positives then all negatives, the two
    category columns through stack_scalars) feeding the REAL NAML and LSTURNewsEncoder: scores, user embeddings and
    LSTUR news vectors per session.  Outputs only -- corpus, weights and sessions regenerate from cases.NAML_DATA."""
    from xnrs.data.dataset import NewsRecDataset
    from xnrs.utils import add_batch_dim_
    c = cases.NAML_DATA
    news_feat, sessions = cases.naml_corpus(c)
    ds = NewsRecDataset(uds=sessions, news_feat=news_feat, mode="eval", n_negatives=None, l_seq=c["S"], l_hist=c["l_hist"],
                        text_features=["title_emb", "abstract_emb"], catg_features=["category_index", "subcategory_index"])
    cfg = Cfg(cases.naml_data_cfg(c))
    naml = load(make_model(cfg), c["model_seed"])
    lcfg = Cfg(dict(cfg, catg_features=["category_index", "subcategory_index"]))
    lstur = load(LSTURNewsEncoder(lcfg), c["model_seed"] + 1)
    out = {}
    for i in range(len(sessions)):
        it = ds[i]
        batch = {"user_features": it["user_features"], "candidate_features": it["candidate_features"]}
        add_batch_dim_(batch)
        with torch.no_grad():
            out[f"naml_ids/s{i}/r"] = npy(naml(batch))
            out[f"naml_ids/s{i}/ue"] = npy(naml.get_user_embeddings(batch))
            hist = batch["user_features"]["history"]
            e, m = lstur(hist["title_emb"], hist["category_index"], hist["subcategory_index"])
            out[f"lstur_ids/s{i}/e"], out[f"lstur_ids/s{i}/m"] = npy(e), npy(m)
    return out


def shipped_config_cases():
    """The reference's make_model on its own shipped YAMLs (config/mind_small_{NRMS,CL,NAML}.yml): the flat cfg keys the
    model constructors read (data: hyper-parameters) and the resulting state_dict key -> shape map + parameter count.
    north_star: "drops into ... the NRMS/NAML/LSTUR configs unchanged".  mind_small_LSTUR.yml is recorded with the
    exception the REFERENCE itself raises when it builds it (SURVEY.md finding 5)."""
    import yaml
    read = ("model", "scoring", "total_emb_dim", "title_emb_dim", "bias", "n_heads", "d_backbone", "p_dropout", "cat_emb_dim",
            "sub_emb_dim", "n_categories", "n_subcategories", "catg_features", "text_features", "user_features", "add_features",
            "hist_len", "seq_len", "n_users", "st_hist_len", "base_model")
    out = {}
    for name in ("mind_small_NRMS", "mind_small_CL", "mind_small_NAML", "mind_small_LSTUR"):
        full = yaml.safe_load(open(f"{REF}/config/{name}.yml"))
        cfg = {k: full[k] for k in read if k in full}
        entry = {"cfg": cfg}
        try:
            model = make_model(Cfg(full))
            entry["state_dict"] = {k: list(v.shape) for k, v in model.state_dict().items()}
            entry["n_params"] = int(sum(p.numel() for p in model.parameters()))
        except Exception as e:  # noqa: BLE001
            entry["reference_error"] = type(e).__name__
        out[name] = entry
    return out


def error_cases():
    """Pin the reference's D % h != 0 failure (layers.py:111,133)."""
    mod = layers.MultiHeadAttention(16, 300)
    try:
        mod(torch.zeros(2, 30, 300), None)
        return {"err": "none"}
    except RuntimeError as e:  # noqa: BLE001
        return {"err": "RuntimeError", "msg": str(e)}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    groups = {
        "blocks": block_cases(),
        "encoders": encoder_cases(),
        "models": model_cases(),
        "lstur": lstur_case(),
        "grads": grad_cases(),
        "data": data_cases(),
        "naml_ids": naml_data_cases(),
        "grads_shipped": grad_shipped_cases(),
        "grads_shipped_standard": grad_shipped_cases(cases.GRAD_SHIPPED_STD, "gss"),
        "grads_shipped_naml": grad_shipped_cases(cases.GRAD_SHIPPED_NAML, "gsn"),
    }
    for g, d in groups.items():
        np.savez_compressed(os.path.join(HERE, f"{g}.npz"), **d)
        print(g, len(d), "arrays", sum(v.nbytes for v in d.values()), "bytes")
    with open(os.path.join(HERE, "shipped_configs.json"), "w") as f:
        json.dump(shipped_config_cases(), f, indent=1, sort_keys=True)
    meta = {"torch": torch.__version__, "numpy": np.__version__, "errors": error_cases()}
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
