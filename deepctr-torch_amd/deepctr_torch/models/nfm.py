# -*- coding: utf-8 -*-
"""NFM (reference models/nfm.py:16-80): linear + DNN over the Bi-Interaction pooling of the embeddings.

Forward = the fused gather, ONE kernel for BiInteractionPooling + the concatenation with the dense features
(csrc/fm.hip: the tower's input row ``[bi | dense]`` is written directly), the MFMA tower."""
import torch.nn as nn

from .basemodel import BaseModel
from ..layers import BiInteractionPooling


class NFM(BaseModel):
    """Same arguments as the reference (models/nfm.py:38-41)."""
    _fused_step_ok = True

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(128, 128),
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=0, init_std=0.0001, seed=1024, bi_dropout=0,
                 dnn_dropout=0, dnn_activation='relu', task='binary', device='cpu', gpus=None):
        super(NFM, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                                  l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                  device=device, gpus=gpus)
        # tower input = [bi-interaction vector (embedding_size) | dense features]
        self._make_tower(self.compute_input_dim(dnn_feature_columns, include_sparse=False) + self.embedding_size,
                         dnn_hidden_units, dnn_activation, l2_reg_dnn, dnn_dropout, False, init_std, device)
        self.bi_pooling = BiInteractionPooling()
        self.bi_dropout = bi_dropout
        if self.bi_dropout > 0:
            self.dropout = nn.Dropout(bi_dropout)
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        gathered, linear_logit, _ = self.fused_inputs(X, want_fm=False, full=True)   # [B, ld]: fields | dense
        if plan.emb_dim <= 0:
            raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
        nf, D, nd = len(plan.deep), plan.emb_dim, len(plan.dense_cols)
        dnn_input = self.bi_pooling.fused(gathered, nf, D, max(plan.dense_off, nf * D), nd)
        if self.bi_dropout and self.training:
            dnn_input = dnn_input.clone()
            dnn_input[:, :D] = self.dropout(dnn_input[:, :D])
        return [linear_logit, self.tower_logit(dnn_input, D + nd)]
