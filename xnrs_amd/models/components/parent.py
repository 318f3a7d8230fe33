"""Drop-in mirror of xnrs/models/components/parent.py::ParentRec."""
import torch
import torch.nn as nn
from typing import Tuple, Optional


class ParentRec(nn.Module):
    """xnrs/models/components/parent.py:8-81: encode history, encode candidates, user vector, score."""

    def __init__(self, news_encoder: nn.Module, user_encoder: nn.Module, rec_model: nn.Module,
                 text_feature: str = 'title_emb'):
        super(ParentRec, self).__init__()
        self.news_encoder = news_encoder
        self.user_encoder = user_encoder
        self.rec_model = rec_model
        self.text_feature = text_feature

    def _forward(self, history: Tuple[torch.Tensor], candidates: Tuple[torch.Tensor],
                 add_user_feats: Optional[Tuple[torch.Tensor]] = None, return_embeddings: bool = False):
        h, hm = self.news_encoder(history)
        c, _ = self.news_encoder(candidates)
        u = self.user_encoder((h, hm), add_user_feats)
        r = self.rec_model(u, c)
        if return_embeddings:
            return r, u, c
        return r

    def forward(self, batch: dict, return_embeddings: bool = False):
        return self._forward(
            history=batch['user_features']['history'][self.text_feature],
            candidates=batch['candidate_features'][self.text_feature],
            add_user_feats=batch['user_features']['other'],
            return_embeddings=return_embeddings
        )

    def _history(self, batch: dict):
        history = batch['user_features']['history'][self.text_feature]
        if isinstance(history, list) and len(history) == 2:
            history = tuple(history)
        return history

    def get_user_embeddings(self, batch: dict) -> torch.Tensor:
        """parent.py:49-81: (B, E) user embedding from the history alone (used by the contrastive loss)."""
        news_emb, news_mask = self.news_encoder(self._history(batch))
        user_emb = self.user_encoder((news_emb, news_mask))
        return user_emb.squeeze(1)

    def forward_ids(self, table_x: torch.Tensor, table_m: torch.Tensor, hist_ids: torch.Tensor,
                    cand_ids: torch.Tensor, return_embeddings: bool = False):
        """Extension: score impressions given as news ids into a device-resident
        token table.  A history id < 0 is an empty slot (all-zero x and m like dataset.py:82-85): point
        such ids at an all-zero table row."""
        h, hm = self.news_encoder.forward_ids(table_x, table_m, hist_ids)
        c, _ = self.news_encoder.forward_ids(table_x, table_m, cand_ids)
        u = self.user_encoder((h, hm), None)
        r = self.rec_model(u, c)
        if return_embeddings:
            return r, u, c
        return r
