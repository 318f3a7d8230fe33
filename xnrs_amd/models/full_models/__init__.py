"""Same import surface as xnrs.models.full_models for the models on the hot path (xnrs_amd/models/assemblies.py)."""
from ..assemblies import NAML, NRMS, NRMS_LF, BaseRec, LSTURNewsEncoder, MeanRec, ParamFreeRec, StandardRec  # noqa: F401
