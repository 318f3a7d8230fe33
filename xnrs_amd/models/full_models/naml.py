"""Mirror of xnrs/models/full_models/naml.py::NAML (multi-view news encoder)."""
import torch
import torch.nn as nn

from ... import ops
from ..components import TextEncoder, layers


class NAML(nn.Module):
    """xnrs/models/full_models/naml.py:7-160: title + abstract additive TextEncoders, category and
    sub-category embedding->Linear, view-level additive attention, additive user attention, dot score.
    The embedding look-up is folded into the Linear's GEMM as a row gather."""

    def __init__(self, cfg, rec_model):
        super(NAML, self).__init__()
        title_pooler = layers.AdditiveAttention(in_features=cfg.d_backbone, hidden_features=256)
        self.title_encoder = TextEncoder(att=None, pooler=title_pooler, p_dropout=cfg.p_dropout,
                                         in_features=cfg.d_backbone, out_features=cfg.title_emb_dim)
        body_pooler = layers.AdditiveAttention(in_features=cfg.d_backbone, hidden_features=256)
        self.body_encoder = TextEncoder(att=None, pooler=body_pooler, p_dropout=cfg.p_dropout,
                                        in_features=cfg.d_backbone, out_features=cfg.title_emb_dim)
        self.cat_embedder = nn.Embedding(num_embeddings=cfg.n_categories + 1, embedding_dim=cfg.cat_emb_dim)
        self.cat_fc = nn.Linear(in_features=cfg.cat_emb_dim, out_features=cfg.total_emb_dim)
        self.subcat_embedder = nn.Embedding(num_embeddings=cfg.n_subcategories + 1, embedding_dim=cfg.sub_emb_dim)
        self.subcat_fc = nn.Linear(in_features=cfg.sub_emb_dim, out_features=cfg.total_emb_dim)
        self.feature_pooler = layers.AdditiveAttention(in_features=cfg.total_emb_dim, hidden_features=256)
        self.user_encoder = layers.AdditiveAttention(in_features=cfg.title_emb_dim, hidden_features=256)
        self.rec_model = rec_model
        self.emb_dim = cfg.total_emb_dim

    def _news_vectors(self, title, abstract, ctg, subctg):
        """naml.py:76-107 for one side (history or candidates) -> ((B,N,E), title mask (B,N,1))."""
        device = next(self.parameters()).device
        t, tm = self.title_encoder(title)
        a, _ = self.body_encoder(abstract)
        b, n, e = t.shape
        ce = ops.embedding_linear(ctg.to(device), self.cat_embedder, self.cat_fc)
        se = ops.embedding_linear(subctg.to(device), self.subcat_embedder, self.subcat_fc)
        views = torch.cat([t, a, ce, se], dim=2).reshape((b * n, 4, self.emb_dim))
        v = self.feature_pooler(views).reshape((b, n, self.emb_dim))
        return v, tm

    def _forward(self, hist_title_features: tuple, hist_abstract_features: tuple, hist_ctg, hist_subctg,
                 cand_title_features: tuple, cand_abstract_features: tuple, cand_ctg, cand_subctg):
        hist, hist_mask = self._news_vectors(hist_title_features, hist_abstract_features, hist_ctg, hist_subctg)
        cand, _ = self._news_vectors(cand_title_features, cand_abstract_features, cand_ctg, cand_subctg)
        urep = self.user_encoder(hist, hist_mask)
        return self.rec_model(urep, cand)

    def get_user_embeddings(self, batch: dict):
        """naml.py:113-147 -> (B, 1, E) (not squeezed, like the reference)."""
        hf = batch['user_features']['history']
        hist, hist_mask = self._news_vectors(hf['title_emb'], hf['abstract_emb'], hf['category_index'],
                                             hf['subcategory_index'])
        return self.user_encoder(hist, hist_mask)

    def forward(self, batch: dict):
        hf, cf = batch['user_features']['history'], batch['candidate_features']
        return self._forward(
            hist_title_features=hf['title_emb'], hist_abstract_features=hf['abstract_emb'],
            hist_ctg=hf['category_index'], hist_subctg=hf['subcategory_index'],
            cand_title_features=cf['title_emb'], cand_abstract_features=cf['abstract_emb'],
            cand_ctg=cf['category_index'], cand_subctg=cf['subcategory_index'])
