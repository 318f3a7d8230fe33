"""Import-path mirror of xnrs.models.make_model."""
from .assemblies import make_model  # noqa: F401
