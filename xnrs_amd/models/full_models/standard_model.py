"""Mirror of xnrs/models/full_models/standard_model.py and base_model.py (additive-only towers)."""
import torch
import torch.nn as nn

from ..components import ParentRec, TextEncoder, UserEncoder, layers


class StandardRec(ParentRec):
    """xnrs/models/full_models/standard_model.py:6-37 (the contrastive-learning bi-encoder: additive
    pooling + MLP head on both towers, bias=cfg.bias)."""

    def __init__(self, cfg, rec_model: nn.Module):
        title_pooler = layers.AdditiveAttention(in_features=cfg.d_backbone, hidden_features=256)
        title_encoder = TextEncoder(att=None, pooler=title_pooler, p_dropout=cfg.p_dropout,
                                    in_features=cfg.d_backbone, out_features=cfg.title_emb_dim, bias=cfg.bias)
        hist_pooler = layers.AdditiveAttention(in_features=cfg.title_emb_dim, hidden_features=256)
        user_encoder = UserEncoder(pooler=hist_pooler, att=None, head=True, p_dropout=cfg.p_dropout,
                                   emb_dim=cfg.title_emb_dim, bias=cfg.bias)
        super(StandardRec, self).__init__(news_encoder=title_encoder, user_encoder=user_encoder, rec_model=rec_model)

    def get_news_embeddings(self, batch: dict, mode: str = 'history') -> torch.Tensor:
        """standard_model.py:73-100."""
        if mode == 'candidate':
            news_input = batch['candidate_features'][self.text_feature]
        elif mode == 'history':
            news_input = batch['user_features']['history'][self.text_feature]
        else:
            raise ValueError("mode must be 'candidate' or 'history'")
        if isinstance(news_input, list) and len(news_input) == 2:
            news_input = tuple(news_input)
        news_emb, _ = self.news_encoder(news_input)
        return news_emb


class BaseRec(ParentRec):
    """xnrs/models/full_models/base_model.py:8-38 (StandardRec without the user head; the news head
    takes bias=cfg.bias)."""

    def __init__(self, cfg, rec_model: nn.Module):
        title_pooler = layers.AdditiveAttention(in_features=cfg.d_backbone, hidden_features=256)
        title_encoder = TextEncoder(att=None, pooler=title_pooler, p_dropout=cfg.p_dropout,
                                    in_features=cfg.d_backbone, out_features=cfg.title_emb_dim, bias=cfg.bias)
        hist_pooler = layers.AdditiveAttention(in_features=cfg.title_emb_dim, hidden_features=256)
        user_encoder = UserEncoder(pooler=hist_pooler, att=None, head=False, p_dropout=cfg.p_dropout,
                                   emb_dim=cfg.title_emb_dim)
        super(BaseRec, self).__init__(news_encoder=title_encoder, user_encoder=user_encoder, rec_model=rec_model)
