"""Import-path mirror of xnrs.models.components.layers (implementations: xnrs_amd/models/blocks.py)."""
from ..blocks import AdditiveAttention, MaskedMean, MultiHeadAttention  # noqa: F401
