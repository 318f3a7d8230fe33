"""Import-path mirror of xnrs.models.components.parent (implementation: xnrs_amd/models/blocks.py)."""
from ..blocks import ParentRec  # noqa: F401
