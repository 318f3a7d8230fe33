"""Model assemblies of the hot path, built from a declarative tower table instead of one hand-wired
constructor per model.  Class names, cfg keys and state_dict keys follow the reference:

  NRMS, NRMS_LF   xnrs/models/full_models/nrms.py:9-79        StandardRec  standard_model.py:6-37 (+:73-100)
  BaseRec         base_model.py:8-38                           MeanRec      mean_model.py:6-31
  ParamFreeRec    param_free_model.py:6-29                     NAML         naml.py:7-160
  LSTURNewsEncoder lstur.py:162-207 (news tower only)          make_model   xnrs/models/make_model.py:15-56

The additive-attention hidden size 256 is hard-coded by the reference assemblies (nrms.py:18,34).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .blocks import AdditiveAttention, DotScoring, MaskedMean, MultiHeadAttention, ParentRec, TextEncoder, UserEncoder

CFG_BIAS = "cfg.bias"  # marker: the tower takes bias=cfg.bias (otherwise the constructor default True)

#            model key : (news tower: att, pooler, head, bias) , (user tower: att, pooler, head, bias)
_TOWERS = {
    "NRMS":      (dict(att=True, pool="additive", head=True, bias=True),      dict(att=True, pool="additive", head=False, bias=True)),
    "NRMS_LF":   (dict(att=True, pool="additive", head=True, bias=True),      dict(att=False, pool="mean", head=False, bias=True)),
    "standard":  (dict(att=False, pool="additive", head=True, bias=CFG_BIAS), dict(att=False, pool="additive", head=True, bias=CFG_BIAS)),
    "base":      (dict(att=False, pool="additive", head=True, bias=CFG_BIAS), dict(att=False, pool="additive", head=False, bias=True)),
    "mean":      (dict(att=False, pool="mean", head=True, bias=CFG_BIAS),     dict(att=False, pool="mean", head=False, bias=CFG_BIAS)),
    "paramfree": (dict(att=False, pool="mean", head=False, bias=True),        dict(att=False, pool="mean", head=False, bias=True)),
}


def _pooler(kind: str, dim: int, hidden: int = 256) -> nn.Module:
    return AdditiveAttention(in_features=dim, hidden_features=hidden) if kind == "additive" else MaskedMean()


def _towers(key: str, cfg) -> Tuple[TextEncoder, UserEncoder]:
    ns, us = _TOWERS[key]
    d, e = cfg.d_backbone, cfg.title_emb_dim
    bias = lambda spec: cfg.bias if spec["bias"] == CFG_BIAS else True  # noqa: E731
    news = TextEncoder(pooler=_pooler(ns["pool"], d), p_dropout=cfg.p_dropout, out_features=e if ns["head"] else d,
                       in_features=d, head=ns["head"], bias=bias(ns),
                       att=MultiHeadAttention(n_heads=cfg.n_heads, d_model=d) if ns["att"] else None)
    user = UserEncoder(pooler=_pooler(us["pool"], e), p_dropout=cfg.p_dropout, emb_dim=e, head=us["head"], bias=bias(us),
                       att=MultiHeadAttention(n_heads=cfg.n_heads, d_model=e) if us["att"] else None)
    return news, user


def _bi_encoder(key: str, doc: str):
    def __init__(self, cfg, rec_model: nn.Module):
        if key == "paramfree":
            assert cfg.title_emb_dim == cfg.d_backbone
        news, user = _towers(key, cfg)
        ParentRec.__init__(self, news_encoder=news, user_encoder=user, rec_model=rec_model)
    return type(doc.split(":")[0], (ParentRec,), {"__init__": __init__, "__doc__": doc})


NRMS = _bi_encoder("NRMS", "NRMS: multi-head self-attention + additive attention on both towers (nrms.py:9-47)")
NRMS_LF = _bi_encoder("NRMS_LF", "NRMS_LF: NRMS news tower + mean-pooling user tower (nrms.py:49-79)")
BaseRec = _bi_encoder("base", "BaseRec: additive towers, MLP head on the news tower only (base_model.py:8-38)")
MeanRec = _bi_encoder("mean", "MeanRec: mean-pooling towers (mean_model.py:6-31)")
ParamFreeRec = _bi_encoder("paramfree", "ParamFreeRec: mean pooling, no parameters (param_free_model.py:6-29)")
_StandardBase = _bi_encoder("standard", "StandardRec: the contrastive-learning bi-encoder (standard_model.py:6-37)")


class StandardRec(_StandardBase):
    __doc__ = _StandardBase.__doc__

    def get_news_embeddings(self, batch: dict, mode: str = 'history') -> torch.Tensor:
        """standard_model.py:73-100: news vectors of the candidates or of the history."""
        src = {'candidate': lambda: batch['candidate_features'], 'history': lambda: batch['user_features']['history']}
        if mode not in src:
            raise ValueError("mode must be 'candidate' or 'history'")
        feats = src[mode]()[self.text_feature]
        return self.news_encoder(tuple(feats) if isinstance(feats, list) else feats)[0]


class NAML(nn.Module):
    """naml.py:7-160: title + abstract additive TextEncoders, category / sub-category embedding -> Linear (one
    GEMM with the table rows gathered by index), view-level additive attention over the four views, additive
    user attention, dot score."""

    def __init__(self, cfg, rec_model):
        super().__init__()
        d, e = cfg.d_backbone, cfg.title_emb_dim
        for name in ("title_encoder", "body_encoder"):
            setattr(self, name, TextEncoder(att=None, pooler=_pooler("additive", d), p_dropout=cfg.p_dropout,
                                            in_features=d, out_features=e))
        self.cat_embedder = nn.Embedding(num_embeddings=cfg.n_categories + 1, embedding_dim=cfg.cat_emb_dim)
        self.cat_fc = nn.Linear(in_features=cfg.cat_emb_dim, out_features=cfg.total_emb_dim)
        self.subcat_embedder = nn.Embedding(num_embeddings=cfg.n_subcategories + 1, embedding_dim=cfg.sub_emb_dim)
        self.subcat_fc = nn.Linear(in_features=cfg.sub_emb_dim, out_features=cfg.total_emb_dim)
        self.feature_pooler = _pooler("additive", cfg.total_emb_dim)
        self.user_encoder = _pooler("additive", e)
        self.rec_model = rec_model
        self.emb_dim = cfg.total_emb_dim

    def _pool_views(self, t, a, cat, sub):
        """naml.py:94-107: concat the four views per news, additive attention over them (no mask)."""
        b, n, _ = t.shape
        stacked = torch.cat([t, a, cat, sub], dim=2).reshape(b * n, 4, self.emb_dim)
        return self.feature_pooler(stacked).reshape(b, n, self.emb_dim)

    def _news_vectors(self, feats: dict):
        """One side (history or candidates), naml.py:76-107 -> ((B,N,E), title mask (B,N,1))."""
        dev = self.cat_fc.weight.device
        t, tm = self.title_encoder(feats['title_emb'])
        a, _ = self.body_encoder(feats['abstract_emb'])
        return self._pool_views(t, a, ops.embedding_linear(feats['category_index'].to(dev), self.cat_embedder, self.cat_fc),
                                ops.embedding_linear(feats['subcategory_index'].to(dev), self.subcat_embedder, self.subcat_fc)), tm

    def get_user_embeddings(self, batch: dict):
        """naml.py:113-147 -> (B,1,E), not squeezed (like the reference)."""
        return self.user_encoder(*self._news_vectors(batch['user_features']['history']))

    def forward(self, batch: dict):
        cand, _ = self._news_vectors(batch['candidate_features'])
        return self.rec_model(self.get_user_embeddings(batch), cand)

    # ---- device-resident data path: the same hooks as ParentRec (blocks.py).  A NAML news needs FOUR things by its row
    # id -- title tokens, abstract tokens, category, sub-category (dataset.py:63-124 gathers all of them by the same news
    # id) -- so the store carries two token tables and two int32 columns; both TextEncoders gather their rows inside
    # their first GEMM's load and the two embedding->Linear views are one row-gathered GEMM each.
    title_feature, body_feature = 'title_emb', 'abstract_emb'
    cat_column, subcat_column = 'category_index', 'subcategory_index'

    def encode_news_ids(self, store, ids: torch.Tensor, dedup: bool = False):
        """News vectors of table rows `ids:(B,N)` -> (vecs:(B,N,E), title mask:(B,N,1)).  dedup=True encodes every
        distinct row once and scatters the vectors back (less algorithmic work: reported separately)."""
        b, n = ids.shape
        flat = ids.reshape(1, -1)
        inv = None
        if dedup:
            uniq, inv = torch.unique(flat.reshape(-1), return_inverse=True)  # index bookkeeping only
            flat = uniq.reshape(1, -1)
        t, tm = self.title_encoder.forward_ids(*store.text(self.title_feature), flat)
        a, _ = self.body_encoder.forward_ids(*store.text(self.body_feature), flat)
        rows = flat.long()
        cat = ops.embedding_linear(store.column(self.cat_column)[rows], self.cat_embedder, self.cat_fc)
        sub = ops.embedding_linear(store.column(self.subcat_column)[rows], self.subcat_embedder, self.subcat_fc)
        v = self._pool_views(t, a, cat, sub)[0]
        tm = tm[0]
        if inv is not None:
            v, tm = v[inv], tm[inv]
        return v.reshape(b, n, self.emb_dim), tm.reshape(b, n, 1)

    def encode_user(self, h: torch.Tensor, hm: torch.Tensor) -> torch.Tensor:
        return self.user_encoder(h, hm)

    def forward_store(self, store, hist_ids: torch.Tensor, cand_ids: torch.Tensor, return_embeddings: bool = False,
                      dedup: bool = False):
        """forward() with the batch given as table rows of a NewsStore (row 0 = the empty history slot: all-zero tokens
        and masks, category 0 -- exactly the padding of dataset.py:82-85 and stack_scalars)."""
        H = hist_ids.shape[1]
        if dedup:
            v, vm = self.encode_news_ids(store, torch.cat([hist_ids, cand_ids], dim=1), dedup=True)
            h, hm, c = v[:, :H].contiguous(), vm[:, :H].contiguous(), v[:, H:].contiguous()
        else:
            h, hm = self.encode_news_ids(store, hist_ids)
            c, _ = self.encode_news_ids(store, cand_ids)
        u = self.encode_user(h, hm)
        r = self.rec_model(u, c)
        return (r, u, c) if return_embeddings else r

    forward_ids = forward_store


class LSTURNewsEncoder(nn.Module):
    """lstur.py:162-207: additive TextEncoder (hidden = title_emb_dim) (+) category [(+) sub-category] embedding,
    concatenated.  (LSTUR's GRU user tower is outside the path and its committed config crashes in the
    reference itself, SURVEY.md finding 5.)"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.title_encoder = TextEncoder(pooler=_pooler("additive", cfg.d_backbone, cfg.title_emb_dim), p_dropout=cfg.p_dropout,
                                         out_features=cfg.title_emb_dim, in_features=cfg.d_backbone, head=True, bias=cfg.bias)
        self.cat_embedder = nn.Embedding(num_embeddings=cfg.n_categories + 1, embedding_dim=cfg.cat_emb_dim)
        if 'subcategory_index' in cfg.catg_features:
            self.subcat_embedder = nn.Embedding(num_embeddings=cfg.n_subcategories + 1, embedding_dim=cfg.cat_emb_dim)

    def _concat(self, emb, cat_idxs, subcat_idxs):
        parts = [emb, self.cat_embedder(cat_idxs.to(emb.device).long())]  # table look-ups + concat: data movement only
        if subcat_idxs is not None:
            assert hasattr(self, 'subcat_embedder')
            parts.append(self.subcat_embedder(subcat_idxs.to(emb.device).long()))
        return torch.cat(parts, dim=2)

    def forward(self, title_features, cat_idxs: torch.Tensor, subcat_idxs: Optional[torch.Tensor]):
        emb, m = self.title_encoder(title_features)
        return self._concat(emb, cat_idxs, subcat_idxs), m

    def forward_ids(self, store, ids: torch.Tensor, dedup: bool = False):
        """forward() by table rows `ids:(B,N)` of a NewsStore: tokens gathered in the first GEMM's load, the category
        columns looked up by the same rows (dataset.py:111-124)."""
        emb, m = self.title_encoder.forward_ids(*store.text('title_emb'), ids, dedup=dedup)
        rows = ids.long()
        sub = store.column('subcategory_index')[rows] if hasattr(self, 'subcat_embedder') else None
        return self._concat(emb, store.column('category_index')[rows], sub), m


# ------------------------------------------------------------------------------------------- factory
_MODELS = {"standard": StandardRec, "base": BaseRec, "mean": MeanRec, "NRMS": NRMS, "NAML": NAML}
# reference models / scorers that are NOT on the path BASELINE.json names: they keep
# running on the reference's own stock-torch classes
_OUT_OF_SCOPE_MODELS = ("smallNAML", "NPA", "LSTUR", "CAUM")
_OUT_OF_SCOPE_SCORING = ("bilin", "fc", "CAUMScoring")


def make_model(cfg):
    """xnrs/models/make_model.py:15-56: same cfg keys (scoring, model, ...), same ValueError on unknown names
    ('nonlin' included: its class does not exist in the reference either, make_model.py:25-26)."""
    if cfg.scoring in _OUT_OF_SCOPE_SCORING or cfg.model in _OUT_OF_SCOPE_MODELS:
        what = f"cfg.scoring={cfg.scoring!r}" if cfg.scoring in _OUT_OF_SCOPE_SCORING else f"cfg.model={cfg.model!r}"
        raise NotImplementedError(f"{what} is outside the MI355X hot path (dot scorer; NRMS / standard / base / mean / "
                                  "NAML); use the reference's torch implementation for it")
    if cfg.scoring != 'dot':
        raise ValueError(f'invalid value for cfg.scoring: {cfg.scoring}')
    if cfg.model not in _MODELS:
        raise ValueError(f'invalid value for cfg.model: {cfg.model}')
    return _MODELS[cfg.model](cfg, DotScoring())
