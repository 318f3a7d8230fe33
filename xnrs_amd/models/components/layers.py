"""Drop-in mirrors of xnrs/models/components/layers.py for the hot-path blocks.

Same class names, constructor signatures, parameter names and registration order (so state_dicts
are interchangeable with the reference), but ``forward`` runs hand-written gfx950 HIP kernels via
libxnrs_hip.so.  There is no torch fallback: CPU tensors raise.
"""
import torch
import torch.nn as nn

from ... import ops


class MaskedMean(nn.Module):
    """xnrs/models/components/layers.py:19-37."""

    def __init__(self):
        super(MaskedMean, self).__init__()

    def forward(self, x: torch.Tensor, m: torch.Tensor):
        '''x: (B, N, D), m: (B, N, 1) -> masked average over dim N, shape (B, 1, D)'''
        return ops.masked_mean(x, m)


class AdditiveAttention(torch.nn.Module):
    """xnrs/models/components/layers.py:40-69 (un-stabilised exp, +1e-8 normaliser)."""

    def __init__(self, in_features, hidden_features):
        super(AdditiveAttention, self).__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, 1)

    def forward(self, x: torch.Tensor, m: torch.Tensor = None, return_weights: bool = False):
        '''x: (B, N, D), m: optional (B, N, 1) -> (B, 1, D) [, weights (B, N, 1)]'''
        return ops.additive_attention(x, m, self, return_weights)


class MultiHeadAttention(nn.Module):
    """xnrs/models/components/layers.py:105-156 (query-ROW mask; Dropout(0.1) on the probabilities
    in train mode)."""

    def __init__(self, n_heads, d_model, dropout=0.1, scaled=True):
        super().__init__()
        self.scaled = scaled
        self.d_model = d_model
        self.d_k = d_model // n_heads
        self.h = n_heads
        # registration order of the reference: q, v, k, dropout, out
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.out = nn.Linear(d_model, d_model)

    def forward(self, x: torch.Tensor, m: torch.Tensor):
        '''x: (B, S, D), m: (B, S, 1) or None -> (B, S, D)'''
        return ops.multi_head_attention(x, m, self)
