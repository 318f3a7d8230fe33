from .make_model import make_model
from .components import TextEncoder, UserEncoder, ParentRec, layers, scoring
from .full_models import NRMS, NRMS_LF, StandardRec, BaseRec, MeanRec, ParamFreeRec, NAML, LSTURNewsEncoder
