"""Pooling over an explicit ``[B, T, D]`` VarLen embedding (reference layers/sequence.py:9-77).

Inside the models this layer is never instantiated: sum / mean / max pooling happens in the gather
kernel (``csrc/embed.hip: pool_field``) while the rows are still in registers.  The class remains for
code that pools tensors it built itself (DIN-style models, user code)."""
import torch
import torch.nn as nn


class SequencePoolingLayer(nn.Module):
    def __init__(self, mode='mean', supports_masking=False, device='cpu'):
        super(SequencePoolingLayer, self).__init__()
        if mode not in ['sum', 'mean', 'max']:
            raise ValueError('parameter mode should in [sum, mean, max]')
        self.supports_masking, self.mode, self.device = supports_masking, mode, device
        self.eps = torch.FloatTensor([1e-8]).to(device)
        self.to(device)

    def _sequence_mask(self, lengths, maxlen=None, dtype=torch.bool):
        if maxlen is None:
            maxlen = lengths.max()
        steps = torch.arange(0, maxlen, 1).to(lengths.device)
        return (steps < torch.unsqueeze(lengths, dim=-1)).type(dtype)

    def forward(self, seq_value_len_list):
        if self.supports_masking:
            seq, mask = seq_value_len_list                    # [B, T, D], [B, T] bool
            mask = mask.float()
            length = torch.sum(mask, dim=-1, keepdim=True)    # [B, 1]
            mask = mask.unsqueeze(2)                          # [B, T, 1]
        else:
            seq, length = seq_value_len_list                  # [B, T, D], [B, 1] int
            mask = self._sequence_mask(length, maxlen=seq.shape[1], dtype=torch.float32)  # [B, 1, T]
            mask = torch.transpose(mask, 1, 2)                # [B, T, 1]
        mask = mask.expand(-1, -1, seq.shape[-1])
        if self.mode == 'max':
            return torch.max(seq - (1 - mask) * 1e9, dim=1, keepdim=True)[0]
        pooled = torch.sum(seq * mask, dim=1, keepdim=False)
        if self.mode == 'mean':
            pooled = torch.div(pooled, length.type(torch.float32) + self.eps.to(length.device))
        return torch.unsqueeze(pooled, dim=1)
