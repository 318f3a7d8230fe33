"""Import-path mirror of xnrs.models.components.scoring (implementation: xnrs_amd/models/blocks.py)."""
from ..blocks import DotScoring  # noqa: F401
