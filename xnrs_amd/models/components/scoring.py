"""Drop-in mirror of xnrs/models/components/scoring.py::DotScoring."""
import torch
import torch.nn as nn

from ... import ops


class DotScoring(nn.Module):
    """xnrs/models/components/scoring.py:6-23."""

    def __init__(self, normalize: bool = False):
        super(DotScoring, self).__init__()
        self.normalize = normalize

    def forward(self, u: torch.Tensor, c: torch.Tensor):
        '''u: (B, 1, D), c: (B, N, D) -> scores (B, N, 1)'''
        return ops.dot_scoring(u, c, self.normalize)
