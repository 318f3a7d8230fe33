"""Import-path mirror of xnrs.models.components.news_encoding (implementation: xnrs_amd/models/blocks.py)."""
from ..blocks import TextEncoder  # noqa: F401
