"""Activations a tower may be built with (the names of reference layers/activation.py:57-84; boundary code on plain
PyTorch).  ``relu`` and ``linear`` towers run on the MFMA tower kernels (csrc/mlp.hip, ``_hip/mlp.tower_layers`` looks for
exactly ``nn.ReLU`` / ``Identity``); every other activation keeps the tower on ``nn.Linear`` + this module."""
import torch
import torch.nn as nn


class Identity(nn.Module):
    """``linear``: what ``_hip/mlp.tower_layers`` recognises as "no activation"."""

    def __init__(self, **kwargs):
        super(Identity, self).__init__()

    def forward(self, inputs):
        return inputs


class Dice(nn.Module):
    """DIN's data-adaptive activation (reference layers/activation.py:6-45): with ``p = sigmoid(BatchNorm(x))`` over the
    feature axis, ``alpha * (1 - p) * x + p * x``.  ``dim`` 2: ``[B, E]``; 3: ``[B, T, E]`` (the feature axis is moved next
    to the batch for ``BatchNorm1d`` and back).  State-dict keys as the reference's: ``bn.*``, ``alpha``."""

    def __init__(self, emb_size, dim=2, epsilon=1e-8, device='cpu'):
        super(Dice, self).__init__()
        if dim not in (2, 3):
            raise AssertionError("Dice takes [B, E] (dim=2) or [B, T, E] (dim=3) inputs")
        self.dim = dim
        self.bn = nn.BatchNorm1d(emb_size, eps=epsilon)
        self.sigmoid = nn.Sigmoid()
        self.alpha = nn.Parameter(torch.zeros((emb_size,) if dim == 2 else (emb_size, 1)).to(device))

    def forward(self, x):
        if x.dim() != self.dim:
            raise AssertionError("Dice(dim=%d) got a %d-dimensional input" % (self.dim, x.dim()))
        h = x.transpose(1, 2) if self.dim == 3 else x           # [B, E(, T)]: BatchNorm1d's channel axis
        gate = self.sigmoid(self.bn(h))
        h = self.alpha * (1 - gate) * h + gate * h
        return h.transpose(1, 2) if self.dim == 3 else h


# name -> factory(hidden_size, dice_dim)
_BY_NAME = {
    'sigmoid': lambda hidden_size, dice_dim: nn.Sigmoid(),
    'linear': lambda hidden_size, dice_dim: Identity(),
    'relu': lambda hidden_size, dice_dim: nn.ReLU(inplace=True),
    'prelu': lambda hidden_size, dice_dim: nn.PReLU(),
}


def _dice(hidden_size, dice_dim):
    assert dice_dim
    return Dice(hidden_size, dice_dim)


_BY_NAME['dice'] = _dice


def activation_layer(act_name, hidden_size=None, dice_dim=2):
    """An activation module from its name (case-insensitive: sigmoid, linear, relu, dice, prelu) or from an ``nn.Module``
    subclass (instantiated without arguments); anything else raises ``NotImplementedError`` like the reference."""
    if isinstance(act_name, str):
        make = _BY_NAME.get(act_name.lower())
        if make is None:
            raise NotImplementedError(act_name)
        return make(hidden_size, dice_dim)
    if isinstance(act_name, type) and issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError
