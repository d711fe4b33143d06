"""Feature-interaction layers of the hot path (reference layers/interaction.py), each backed by a
hand-written gfx950 kernel in ``libdctr_hip.so``.  Constructor signatures, input/output shapes,
parameter names and error behaviour follow the reference so models and checkpoints drop in."""
import itertools

import torch
import torch.nn as nn

from .._hip import ops as _ops

__all__ = ["FM"]


class FM(nn.Module):
    """Pairwise (order-2) interactions without linear term and bias:
    ``0.5 * sum_d((sum_f e)^2 - sum_f e^2)`` -- ``[B, F, D] -> [B, 1]`` (reference interaction.py:12-34).
    Kernel: ``dctr_fm_fwd`` / ``dctr_fm_bwd`` (csrc/fm.hip)."""

    def __init__(self):
        super(FM, self).__init__()

    def forward(self, inputs):
        return _ops.FMFunction.apply(inputs)
