"""Layer zoo of the hot path.  ★ interaction layers call hand-written gfx950 kernels through
``deepctr_torch._hip``; the MLP tower / activations are the boundary and stay ordinary PyTorch-ROCm
modules."""
from .activation import Dice, Identity, activation_layer
from .core import DNN, PredictionLayer
from .interaction import *  # noqa: F401,F403
from .sequence import SequencePoolingLayer
from .utils import concat_fun, slice_arrays
