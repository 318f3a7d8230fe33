"""Mirrors of xnrs/models/full_models/mean_model.py and param_free_model.py (MaskedMean towers)."""
import torch.nn as nn

from ..components import ParentRec, TextEncoder, UserEncoder, layers


class MeanRec(ParentRec):
    """xnrs/models/full_models/mean_model.py:6-31."""

    def __init__(self, cfg, rec_model: nn.Module):
        title_encoder = TextEncoder(att=None, pooler=layers.MaskedMean(), p_dropout=cfg.p_dropout,
                                    in_features=cfg.d_backbone, out_features=cfg.title_emb_dim, bias=cfg.bias)
        user_encoder = UserEncoder(pooler=layers.MaskedMean(), att=None, head=False, p_dropout=cfg.p_dropout,
                                   emb_dim=cfg.title_emb_dim, bias=cfg.bias)
        super(MeanRec, self).__init__(news_encoder=title_encoder, user_encoder=user_encoder, rec_model=rec_model)


class ParamFreeRec(ParentRec):
    """xnrs/models/full_models/param_free_model.py:6-29."""

    def __init__(self, cfg, rec_model: nn.Module):
        assert cfg.title_emb_dim == cfg.d_backbone
        title_encoder = TextEncoder(att=None, head=False, pooler=layers.MaskedMean(), p_dropout=cfg.p_dropout,
                                    out_features=cfg.d_backbone)
        user_encoder = UserEncoder(att=None, head=False, pooler=layers.MaskedMean(), p_dropout=cfg.p_dropout,
                                   emb_dim=cfg.title_emb_dim)
        super(ParamFreeRec, self).__init__(news_encoder=title_encoder, user_encoder=user_encoder, rec_model=rec_model)
