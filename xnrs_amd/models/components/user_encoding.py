"""Import-path mirror of xnrs.models.components.user_encoding (implementation: xnrs_amd/models/blocks.py)."""
from ..blocks import UserEncoder  # noqa: F401
