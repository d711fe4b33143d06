# -*- coding: utf-8 -*-
"""FiBiNET (reference models/fibinet.py:17-102): SENET + bilinear interactions feeding a DNN, plus the linear part.
Device phases: fused gather -> SENET kernel -> ONE bilinear kernel that writes the [B, F(F-1)D + dense] DNN input
(the reference needs ~2 800 ATen launches and a 650-way cat for the same tensor) -> MLP."""
import torch.nn as nn

from .basemodel import BaseModel
from ..inputs import DenseFeat, SparseFeat, VarLenSparseFeat
from .._hip import ops as _ops
from ..layers import DNN, BilinearInteraction, SENETLayer


class FiBiNET(BaseModel):
    """Same arguments as the reference (models/fibinet.py:39-42)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, bilinear_type='interaction',
                 reduction_ratio=3, dnn_hidden_units=(128, 128), l2_reg_linear=1e-5,
                 l2_reg_embedding=1e-5, l2_reg_dnn=0, init_std=0.0001, seed=1024, dnn_dropout=0, dnn_activation='relu',
                 task='binary', device='cpu', gpus=None):
        super(FiBiNET, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                                      l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                      device=device, gpus=gpus)
        self.linear_feature_columns = linear_feature_columns
        self.dnn_feature_columns = dnn_feature_columns
        self.field_size = len(self.embedding_dict)
        self.SE = SENETLayer(self.field_size, reduction_ratio, seed, device)
        self.Bilinear = BilinearInteraction(self.field_size, self.embedding_size, bilinear_type, seed, device)
        self.dnn = DNN(self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, activation=dnn_activation,
                       l2_reg=l2_reg_dnn, dropout_rate=dnn_dropout, use_bn=False, init_std=init_std, device=device)
        self.dnn_linear = nn.Linear(dnn_hidden_units[-1], 1, bias=False).to(device)   # (no reg groups: fibinet.py:60-64)

    def compute_input_dim(self, feature_columns, include_sparse=True, include_dense=True):
        emb_cols = [c for c in feature_columns if isinstance(c, (SparseFeat, VarLenSparseFeat))] \
            if len(feature_columns) else []
        dense_cols = [c for c in feature_columns if isinstance(c, DenseFeat)] if len(feature_columns) else []
        field_size = len(emb_cols)
        dense_input_dim = sum(c.dimension for c in dense_cols)
        sparse_input_dim = field_size * (field_size - 1) * emb_cols[0].embedding_dim
        return (sparse_input_dim if include_sparse else 0) + (dense_input_dim if include_dense else 0)

    def logit_parts(self, X):
        plan = self.model_plan()
        gathered, linear_logit, _ = self.fused_inputs(X, want_fm=False, full=True)
        emb, dense = _ops.split_gathered(gathered, plan)                      # views of the gather's output
        # (lazy: on the GPU pairs + first tower layer become one autograd node -- the gradient of the product slab is made
        # where it is consumed, csrc/bilinear_wide.hip; the first layer's forward stays one library GEMM)
        dnn_input = self.Bilinear.fused_pair(emb, self.SE(emb), dense, lazy=True)
        dnn_logit = self.tower_logit(dnn_input)     # the layers behind the first on csrc/mlp.hip
        if len(self.linear_feature_columns) > 0 and len(self.dnn_feature_columns) > 0:
            return [linear_logit, dnn_logit]
        elif len(self.linear_feature_columns) == 0:
            return [dnn_logit]
        elif len(self.dnn_feature_columns) == 0:
            return [linear_logit]
        raise NotImplementedError
