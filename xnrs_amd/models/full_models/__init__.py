from .standard_model import StandardRec, BaseRec
from .mean_model import MeanRec, ParamFreeRec
from .nrms import NRMS, NRMS_LF
from .naml import NAML
from .lstur import LSTURNewsEncoder
