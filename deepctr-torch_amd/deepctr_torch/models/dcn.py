# -*- coding: utf-8 -*-
"""DCN / DCN-M (reference models/dcn.py:20-96): cross network + DNN over the fused gather's output."""
import torch

from .basemodel import BaseModel
from ..layers import CrossNet


class DCN(BaseModel):
    """Same arguments as the reference (models/dcn.py:44-47).  Like the reference, ``l2_reg_linear`` is NOT forwarded
    to the linear part (``BaseModel`` keeps its own default there, dcn.py:49-51); it regularises ``dnn_linear``."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, cross_num=2, cross_parameterization='vector',
                 dnn_hidden_units=(128, 128), l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_cross=0.00001,
                 l2_reg_dnn=0, init_std=0.0001, seed=1024, dnn_dropout=0, dnn_activation='relu', dnn_use_bn=False,
                 task='binary', device='cpu', gpus=None):
        super(DCN, self).__init__(linear_feature_columns=linear_feature_columns,
                                  dnn_feature_columns=dnn_feature_columns, l2_reg_embedding=l2_reg_embedding,
                                  init_std=init_std, seed=seed, task=task, device=device, gpus=gpus)
        self.dnn_hidden_units = dnn_hidden_units
        self.cross_num = cross_num
        width = self.compute_input_dim(dnn_feature_columns)
        deep, cross = len(dnn_hidden_units) > 0, cross_num > 0
        # dnn_linear reads [cross output (width) | tower output]: whichever of the two exist
        head_in = (width if cross else 0) + (dnn_hidden_units[-1] if deep else 0)
        self._make_tower(width, dnn_hidden_units, dnn_activation, l2_reg_dnn, dnn_dropout, dnn_use_bn, init_std, device,
                         head_in=head_in, l2_head=l2_reg_linear)
        self.crossnet = CrossNet(in_features=self.compute_input_dim(dnn_feature_columns),
                                 layer_num=cross_num, parameterization=cross_parameterization, device=device)
        self.add_regularization_weight(self.crossnet.kernels, l2=l2_reg_cross)
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        full, logit, _ = self.fused_inputs(X, want_fm=False, full=True)
        dnn_input = full[:, :plan.width]
        parts = [logit]
        if len(self.dnn_hidden_units) > 0 and self.cross_num > 0:      # Deep & Cross
            stack_out = torch.cat((self.crossnet(dnn_input), self.tower_hidden(full, plan.width)), dim=-1)
            parts.append(self.dnn_linear(stack_out))
        elif len(self.dnn_hidden_units) > 0:                           # only Deep
            parts.append(self.tower_logit(full, plan.width))
        elif self.cross_num > 0:                                       # only Cross
            parts.append(self.dnn_linear(self.crossnet(dnn_input)))
        return parts
