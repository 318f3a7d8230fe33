"""Drop-in mirror of xnrs/models/components/user_encoding.py::UserEncoder."""
import torch
import torch.nn as nn
from typing import Optional

from ... import ops


class UserEncoder(nn.Module):
    """xnrs/models/components/user_encoding.py:6-81 (head = Linear(E,E)-ReLU-Linear(E,E) as committed
    there at :26-34; ``out_dim`` is accepted and ignored exactly like the reference)."""

    def __init__(self,
                 pooler: nn.Module,
                 p_dropout: float,
                 emb_dim: Optional[int] = None,
                 out_dim: Optional[int] = None,
                 att: Optional[nn.Module] = None,
                 head: bool = False,
                 activation: nn.Module = nn.ReLU(),
                 bias: bool = True):
        super(UserEncoder, self).__init__()
        self.dummy_param = nn.Parameter(torch.zeros(1))
        self.dropout = nn.Dropout(p=p_dropout)
        self.att = att
        self.pooler = pooler
        if head:
            assert emb_dim is not None
            if not isinstance(activation, nn.ReLU):
                raise NotImplementedError('the HIP head kernel implements the reference default ReLU only')
            self.head = nn.Sequential(
                nn.Linear(emb_dim, emb_dim, bias=bias),
                activation,
                nn.Linear(emb_dim, emb_dim, bias=bias)
            )

    def forward(self, inpt: tuple, add_features: Optional[dict] = None, return_weights: bool = False):
        '''inpt = (x: (B, N, D), m: (B, N, 1)) -> (B, 1, D) [, pooling weights (B, N, 1)]'''
        x, m = inpt
        device = next(self.parameters()).device
        x = x.to(device)
        m = m.to(device)
        x = self.dropout(x)
        return ops.user_encoder(x, m, self, return_weights)
