"""MLP tower and prediction head (boundary: PyTorch-ROCm ``nn.Linear`` -> hipBLASLt fp32).
Same parameters / ``state_dict`` keys as the reference (layers/core.py:67-160)."""
import torch
import torch.nn as nn

from .activation import activation_layer


class DNN(nn.Module):
    """``hidden_units`` fully connected layers: Linear -> [BatchNorm] -> activation -> Dropout
    (reference layers/core.py:67-134).  Weights N(0, init_std); biases keep nn.Linear's default."""

    def __init__(self, inputs_dim, hidden_units, activation='relu', l2_reg=0, dropout_rate=0, use_bn=False,
                 init_std=0.0001, dice_dim=3, seed=1024, device='cpu'):
        super(DNN, self).__init__()
        if len(hidden_units) == 0:
            raise ValueError("hidden_units is empty!!")
        self.dropout_rate, self.seed, self.l2_reg, self.use_bn = dropout_rate, seed, l2_reg, use_bn
        self.dropout = nn.Dropout(dropout_rate)
        widths = [inputs_dim] + list(hidden_units)
        self.linears = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))
        if use_bn:
            self.bn = nn.ModuleList(nn.BatchNorm1d(b) for b in widths[1:])
        self.activation_layers = nn.ModuleList(activation_layer(activation, b, dice_dim) for b in widths[1:])
        for name, tensor in self.linears.named_parameters():
            if 'weight' in name:
                nn.init.normal_(tensor, mean=0, std=init_std)
        self.to(device)

    def forward(self, inputs):
        h = inputs
        for i, fc in enumerate(self.linears):
            h = fc(h)
            if self.use_bn:
                h = self.bn[i](h)
            h = self.dropout(self.activation_layers[i](h))
        return h


class PredictionLayer(nn.Module):
    """``sigmoid(logit + bias)`` for task='binary', ``logit + bias`` otherwise
    (reference layers/core.py:137-160)."""

    def __init__(self, task='binary', use_bias=True, **kwargs):
        if task not in ["binary", "multiclass", "regression"]:
            raise ValueError("task must be binary,multiclass or regression")
        super(PredictionLayer, self).__init__()
        self.use_bias, self.task = use_bias, task
        if use_bias:
            self.bias = nn.Parameter(torch.zeros((1,)))

    def forward(self, X):
        out = X + self.bias if self.use_bias else X
        return torch.sigmoid(out) if self.task == "binary" else out
