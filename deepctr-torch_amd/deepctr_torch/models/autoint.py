# -*- coding: utf-8 -*-
"""AutoInt (reference models/autoint.py:17-112): stacked multi-head self-attention over the field embeddings, optionally
beside a DNN.  The gather, the tower and the table update are this package's kernels; the attention layers run as
batched GEMMs on PyTorch-ROCm (layers.InteractingLayer)."""
import torch
import torch.nn as nn

from .basemodel import BaseModel
from ..layers import InteractingLayer


class AutoInt(BaseModel):
    """Same arguments as the reference (models/autoint.py:39-42)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, att_layer_num=3, att_head_num=2, att_res=True,
                 dnn_hidden_units=(256, 128), dnn_activation='relu', l2_reg_dnn=0, l2_reg_embedding=1e-5,
                 dnn_use_bn=False, dnn_dropout=0, init_std=0.0001, seed=1024, task='binary', device='cpu', gpus=None):
        super(AutoInt, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=0,
                                      l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                      device=device, gpus=gpus)
        if len(dnn_hidden_units) <= 0 and att_layer_num <= 0:
            raise ValueError("Either hidden_layer or att_layer_num must > 0")
        self.use_dnn = len(dnn_feature_columns) > 0 and len(dnn_hidden_units) > 0
        field_num = len(self.embedding_dict)
        embedding_size = self.embedding_size
        att_width = field_num * embedding_size if att_layer_num > 0 else 0
        head_in = att_width + (dnn_hidden_units[-1] if len(dnn_hidden_units) > 0 else 0)   # [attention output | tower output]
        self.dnn_hidden_units = dnn_hidden_units
        self.att_layer_num = att_layer_num
        if self.use_dnn:
            self._make_tower(self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, dnn_activation, l2_reg_dnn,
                             dnn_dropout, dnn_use_bn, init_std, device, head_in=head_in, l2_head=False, head_first=True)
        else:
            self.dnn_linear = nn.Linear(head_in, 1, bias=False).to(device)
        self.int_layers = nn.ModuleList(
            [InteractingLayer(embedding_size, att_head_num, att_res, device=device) for _ in range(att_layer_num)])
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        full, logit, _ = self.fused_inputs(X, want_fm=False, full=True)
        parts = [logit]
        att_output = None
        if self.att_layer_num > 0:
            if plan.emb_dim <= 0:
                raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
            att = full[:, :plan.emb_width].reshape(X.shape[0], len(plan.deep), plan.emb_dim)
            for layer in self.int_layers:
                att = layer(att)
            att_output = torch.flatten(att, start_dim=1)
        if len(self.dnn_hidden_units) > 0 and self.att_layer_num > 0:      # Deep & Interacting Layer
            stack_out = torch.cat((att_output, self.tower_hidden(full, plan.width)), dim=-1)
            parts.append(self.dnn_linear(stack_out))
        elif len(self.dnn_hidden_units) > 0:                               # only Deep
            parts.append(self.tower_logit(full, plan.width))
        elif self.att_layer_num > 0:                                       # only Interacting Layer
            parts.append(self.dnn_linear(att_output))
        return parts
