"""Activations of the MLP tower (boundary code: plain PyTorch).  Mirrors layers/activation.py:6-88."""
import torch
import torch.nn as nn


class Dice(nn.Module):
    """Data-adaptive activation of DIN: ``p(x) * x + (1 - p(x)) * alpha * x`` with
    ``p = sigmoid(BatchNorm(x))`` (reference layers/activation.py:6-45)."""

    def __init__(self, emb_size, dim=2, epsilon=1e-8, device='cpu'):
        super(Dice, self).__init__()
        assert dim == 2 or dim == 3
        self.bn = nn.BatchNorm1d(emb_size, eps=epsilon)
        self.sigmoid = nn.Sigmoid()
        self.dim = dim
        shape = (emb_size,) if dim == 2 else (emb_size, 1)
        self.alpha = nn.Parameter(torch.zeros(shape).to(device))

    def forward(self, x):
        assert x.dim() == self.dim
        if self.dim == 3:
            x = torch.transpose(x, 1, 2)
        p = self.sigmoid(self.bn(x))
        out = self.alpha * (1 - p) * x + p * x
        if self.dim == 3:
            out = torch.transpose(out, 1, 2)
        return out


class Identity(nn.Module):
    def __init__(self, **kwargs):
        super(Identity, self).__init__()

    def forward(self, inputs):
        return inputs


def activation_layer(act_name, hidden_size=None, dice_dim=2):
    """Name (or nn.Module subclass) -> activation module (reference layers/activation.py:57-84)."""
    if isinstance(act_name, str):
        key = act_name.lower()
        if key == 'sigmoid':
            return nn.Sigmoid()
        if key == 'linear':
            return Identity()
        if key == 'relu':
            return nn.ReLU(inplace=True)
        if key == 'dice':
            assert dice_dim
            return Dice(hidden_size, dice_dim)
        if key == 'prelu':
            return nn.PReLU()
        raise NotImplementedError(act_name)
    if isinstance(act_name, type) and issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError
