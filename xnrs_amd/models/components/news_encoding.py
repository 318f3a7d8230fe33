"""Drop-in mirror of xnrs/models/components/news_encoding.py::TextEncoder."""
import torch
import torch.nn as nn
from typing import Optional

from ... import ops


class TextEncoder(nn.Module):
    """xnrs/models/components/news_encoding.py:8-60.

    The whole forward (optional self-attention -> pooler -> optional MLP head -> news mask) is one call
    into the HIP pipeline.  Extension next to the unchanged ``forward((x, m))`` API:
    ``forward_ids(table_x, table_m, ids)`` encodes news gathered by id from a device-resident token
    table."""

    def __init__(self, pooler: nn.Module,
                 p_dropout: float,
                 out_features: int,
                 in_features: Optional[int] = 768,
                 head: bool = True,
                 activation: nn.Module = nn.ReLU(),
                 att: Optional[nn.Module] = None,
                 bias: bool = True):
        super(TextEncoder, self).__init__()
        # to make sure the model has at least one param
        self.dummy_param = nn.Parameter(torch.zeros(1))
        self.dropout = nn.Dropout(p=p_dropout)
        self.att = att
        self.pooler = pooler
        if head:
            assert in_features is not None, 'in_features is required if head is True'
            if not isinstance(activation, nn.ReLU):
                raise NotImplementedError('the HIP head kernel implements the reference default ReLU only')
            self.head = nn.Sequential(
                nn.Linear(in_features, out_features, bias=bias),
                activation,
                nn.Linear(out_features, out_features, bias=bias)
            )
        self.out_dim = out_features

    def forward(self, inpt: tuple):
        '''inpt = (x: (B, N, S, D), m: (B, N, S, 1)) -> (y: (B, N, out_dim), news mask (B, N, 1))'''
        x, m = inpt
        device = next(self.parameters()).device
        x = x.to(device)
        m = m.to(device)
        b, n, s, d = x.shape
        x = x.reshape((b * n, s, d))
        m = m.reshape((b * n, s, 1))
        x = self.dropout(x)
        y, hm = ops.text_encoder(x, m, self)
        return y.reshape((b, n, self.out_dim)), hm.reshape((b, n, 1))

    def forward_ids(self, table_x: torch.Tensor, table_m: torch.Tensor, ids: torch.Tensor):
        '''table_x: (n_table, S, D), table_m: (n_table, S[, 1]) resident on the device; ids: (B, N) int
        -> same outputs as forward((table_x[ids], table_m[ids]))'''
        b, n = ids.shape
        y, hm = ops.text_encoder(table_x, table_m, self, ids=ids.reshape(-1))
        return y.reshape((b, n, self.out_dim)), hm.reshape((b, n, 1))
