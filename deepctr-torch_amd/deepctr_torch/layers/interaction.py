"""Feature-interaction layers of the hot path (reference layers/interaction.py), each backed by a
hand-written gfx950 kernel in ``libdctr_hip.so``.  Constructor signatures, input/output shapes,
parameter names and error behaviour follow the reference so models and checkpoints drop in."""
import itertools

import torch
import torch.nn as nn

from .._hip import ops as _ops
from .activation import activation_layer

__all__ = ["FM", "CIN"]


class FM(nn.Module):
    """Pairwise (order-2) interactions without linear term and bias:
    ``0.5 * sum_d((sum_f e)^2 - sum_f e^2)`` -- ``[B, F, D] -> [B, 1]`` (reference interaction.py:12-34).
    Kernel: ``dctr_fm_fwd`` / ``dctr_fm_bwd`` (csrc/fm.hip)."""

    def __init__(self):
        super(FM, self).__init__()

    def forward(self, inputs):
        return _ops.FMFunction.apply(inputs)


class CIN(nn.Module):
    """Compressed Interaction Network of xDeepFM: ``[B, F, D] -> [B, featuremap_num]`` (reference
    interaction.py:159-248; same constructor, same ``conv1ds.<k>.weight [O, h*F, 1]`` / ``bias`` parameters).

    Each layer is ONE fp32-MFMA kernel (``csrc/cin.hip``) that never materialises the reference's
    ``[B, h*F, D]`` outer product (436 MB at the Criteo shape); relu (the default) is fused, any other activation
    module is applied to the kernel's linear output."""

    def __init__(self, field_size, layer_size=(128, 128), activation='relu', split_half=True, l2_reg=1e-5, seed=1024,
                 device='cpu'):
        super(CIN, self).__init__()
        if len(layer_size) == 0:
            raise ValueError("layer_size must be a list(tuple) of length greater than 1")
        self.layer_size = layer_size
        self.field_nums = [field_size]
        self.split_half = split_half
        self.activation = activation_layer(activation)
        self.l2_reg = l2_reg
        self.seed = seed
        self.conv1ds = nn.ModuleList()
        for i, size in enumerate(self.layer_size):
            self.conv1ds.append(nn.Conv1d(self.field_nums[-1] * self.field_nums[0], size, 1))
            if self.split_half:
                if i != len(self.layer_size) - 1 and size % 2 > 0:
                    raise ValueError("layer_size must be even number except for the last layer when split_half=True")
                self.field_nums.append(size // 2)
            else:
                self.field_nums.append(size)
        self.to(device)

    def forward(self, inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))
        x0 = inputs
        hidden = x0
        fused_relu = isinstance(self.activation, nn.ReLU)
        final_result = []
        for i, size in enumerate(self.layer_size):
            conv = self.conv1ds[i]
            curr_out = _ops.CINLayerFunction.apply(hidden, x0, conv.weight.squeeze(-1), conv.bias, fused_relu)
            if not fused_relu and self.activation is not None:
                curr_out = self.activation(curr_out)
            if self.split_half:
                if i != len(self.layer_size) - 1:
                    hidden, direct_connect = torch.split(curr_out, 2 * [size // 2], 1)
                else:
                    direct_connect, hidden = curr_out, None
            else:
                direct_connect, hidden = curr_out, curr_out
            final_result.append(direct_connect)
        return torch.sum(torch.cat(final_result, dim=1), -1)
