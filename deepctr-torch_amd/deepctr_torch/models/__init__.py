"""Drop-in model classes of the MI355X hot path.  Constructor signatures
and ``state_dict`` keys are the reference's (``deepctr_torch/models/*.py``)."""
from .afm import AFM
from .autoint import AutoInt
from .basemodel import BaseModel, Linear
from .dcn import DCN
from .dcnmix import DCNMix
from .deepfm import DeepFM
from .fibinet import FiBiNET
from .nfm import NFM
from .pnn import PNN
from .wdl import WDL
from .xdeepfm import xDeepFM

__all__ = ["BaseModel", "Linear", "DeepFM", "xDeepFM", "FiBiNET", "DCN", "PNN", "NFM", "AFM", "WDL", "AutoInt", "DCNMix"]
