"""Drop-in for `xnrs.models`: same factory, same class names, HIP kernels underneath."""
from .assemblies import (NAML, NRMS, NRMS_LF, BaseRec, LSTURNewsEncoder, MeanRec, ParamFreeRec, StandardRec,  # noqa: F401
                         make_model)
from .blocks import ParentRec, TextEncoder, UserEncoder  # noqa: F401
from .components import layers, scoring  # noqa: F401
