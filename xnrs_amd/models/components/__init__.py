"""Same import surface as xnrs.models.components."""
from ..blocks import ParentRec, TextEncoder, UserEncoder  # noqa: F401
from . import layers, news_encoding, parent, scoring, user_encoding  # noqa: F401
