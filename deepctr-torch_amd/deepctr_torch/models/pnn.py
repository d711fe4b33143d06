# -*- coding: utf-8 -*-
"""PNN (reference models/pnn.py:17-109) with the inner- and outer-product layers on the gfx950 kernels of
csrc/pairwise.hip."""
import torch

from .basemodel import BaseModel
from ..layers import InnerProductLayer, OutterProductLayer


class PNN(BaseModel):
    _fused_step_ok = True
    """Same arguments as the reference (models/pnn.py:38-40)."""

    def __init__(self, dnn_feature_columns, dnn_hidden_units=(128, 128), l2_reg_embedding=1e-5, l2_reg_dnn=0,
                 init_std=0.0001, seed=1024, dnn_dropout=0, dnn_activation='relu', use_inner=True, use_outter=False,
                 kernel_type='mat', task='binary', device='cpu', gpus=None):
        super(PNN, self).__init__([], dnn_feature_columns, l2_reg_linear=0, l2_reg_embedding=l2_reg_embedding,
                                  init_std=init_std, seed=seed, task=task, device=device, gpus=gpus)
        if kernel_type not in ['mat', 'vec', 'num']:
            raise ValueError("kernel_type must be mat,vec or num")
        self.use_inner = use_inner
        self.use_outter = use_outter
        self.kernel_type = kernel_type
        self.task = task
        product_out_dim = 0
        num_inputs = self.compute_input_dim(dnn_feature_columns, include_dense=False, feature_group=True)
        num_pairs = int(num_inputs * (num_inputs - 1) / 2)
        if self.use_inner:
            product_out_dim += num_pairs
            self.innerproduct = InnerProductLayer(device=device)
        if self.use_outter:
            product_out_dim += num_pairs
            self.outterproduct = OutterProductLayer(num_inputs, self.embedding_size, kernel_type=kernel_type,
                                                    device=device)
        self._make_tower(product_out_dim + self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, dnn_activation,
                         l2_reg_dnn, dnn_dropout, False, init_std, device)
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        gathered, _, _ = self.fused_inputs(X, want_fm=False)      # [B, F*D | dense]
        B, nf = X.shape[0], len(plan.deep)
        parts = [gathered[:, :plan.emb_width]]
        if self.use_inner or self.use_outter:
            if plan.emb_dim <= 0:
                raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
            emb = gathered[:, :plan.emb_width].reshape(B, nf, plan.emb_dim)
        if self.use_inner:
            parts.append(torch.flatten(self.innerproduct(emb), start_dim=1))
        if self.use_outter:
            parts.append(self.outterproduct(emb))
        if plan.dense_cols:
            parts.append(gathered[:, plan.emb_width:])
        dnn_input = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        return [self.tower_logit(dnn_input)]
