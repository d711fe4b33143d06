"""``SequencePoolingLayer``: sum / mean / max over the valid positions of an explicit ``[B, T, D]`` VarLen embedding
(the layer of reference layers/sequence.py:9-77, for code that pools tensors it built itself:
``inputs.get_varlen_pooling_list``, DIN-style user models).

The models of this package never instantiate it: their pooling happens inside the gather kernel
(``csrc/embed.hip: pool_field``) while the rows are still in registers.  This module states the same three reductions
with the kernel's conventions on a tensor: a position is valid when its mask bit is set (``supports_masking``) or its
index is below the row's length; ``mean`` divides by ``count + 1e-8``; ``max`` lowers every padded position by 1e9
before the reduction, so a row without valid positions yields its values minus 1e9, as the reference's does."""
import torch
import torch.nn as nn

_MODES = ('sum', 'mean', 'max')
_PAD_DROP = 1e9


def _valid_positions(lengths, T):
    """``[B, T]`` bool: position t of row b is valid when t < lengths[b] (lengths ``[B]`` or ``[B, 1]``)."""
    return torch.arange(T, device=lengths.device).unsqueeze(0) < lengths.reshape(-1, 1)


class SequencePoolingLayer(nn.Module):
    """``forward([seq [B, T, D], mask [B, T] bool])`` with ``supports_masking`` else ``forward([seq, lengths [B, 1]])``
    -> ``[B, 1, D]``."""

    def __init__(self, mode='mean', supports_masking=False, device='cpu'):
        super(SequencePoolingLayer, self).__init__()
        if mode not in _MODES:
            raise ValueError('parameter mode should in [sum, mean, max]')
        self.mode, self.supports_masking, self.device = mode, supports_masking, device
        self.eps = torch.FloatTensor([1e-8]).to(device)
        self.to(device)

    def forward(self, seq_value_len_list):
        seq, second = seq_value_len_list
        if self.supports_masking:
            valid = second.to(torch.bool)
            count = valid.sum(dim=1, keepdim=True).to(torch.float32)
        else:
            valid = _valid_positions(second, seq.shape[1])
            count = second.reshape(-1, 1).to(torch.float32)
        keep = valid.unsqueeze(-1).to(seq.dtype)                         # [B, T, 1], broadcast over D
        if self.mode == 'max':
            return (seq - (1.0 - keep) * _PAD_DROP).amax(dim=1, keepdim=True)
        total = (seq * keep).sum(dim=1, keepdim=True)                    # [B, 1, D]
        if self.mode == 'mean':
            total = total / (count + self.eps.to(count.device)).unsqueeze(-1)
        return total
