"""Mirror of the in-scope part of xnrs/models/full_models/lstur.py: the LSTUR *news* encoder.

The GRU user encoder (lstur.py:83-159) is out of scope: it is not on the path BASELINE.json names and
the committed LSTUR config crashes in the reference itself (SURVEY.md finding 5)."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ..components import layers, news_encoding


class LSTURNewsEncoder(nn.Module):
    """xnrs/models/full_models/lstur.py:162-207: additive TextEncoder (+) category [(+) sub-category]
    embedding, concatenated."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        title_pooler = layers.AdditiveAttention(in_features=cfg.d_backbone, hidden_features=cfg.title_emb_dim)
        self.title_encoder = news_encoding.TextEncoder(
            pooler=title_pooler, p_dropout=cfg.p_dropout, out_features=cfg.title_emb_dim,
            in_features=cfg.d_backbone, head=True, bias=cfg.bias)
        self.cat_embedder = nn.Embedding(num_embeddings=cfg.n_categories + 1, embedding_dim=cfg.cat_emb_dim)
        if 'subcategory_index' in cfg.catg_features:
            self.subcat_embedder = nn.Embedding(num_embeddings=cfg.n_subcategories + 1,
                                                embedding_dim=cfg.cat_emb_dim)

    def forward(self, title_features: Tuple[torch.tensor], cat_idxs: torch.tensor,
                subcat_idxs: Optional[torch.tensor]):
        title_emb, m = self.title_encoder(title_features)
        # plain table look-ups + concat: data movement only, no arithmetic
        cat_emb = self.cat_embedder(cat_idxs.to(title_emb.device).long())
        emb = torch.cat([title_emb, cat_emb], dim=2)
        if subcat_idxs is not None:
            assert hasattr(self, 'subcat_embedder')
            emb = torch.cat([emb, self.subcat_embedder(subcat_idxs.to(title_emb.device).long())], dim=2)
        return emb, m
