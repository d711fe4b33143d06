# -*- coding: utf-8 -*-
"""xDeepFM (reference models/xdeepfm.py:17-107): linear + CIN + DNN over shared embeddings.
One fused gather feeds both towers; each CIN layer is one fp32-MFMA kernel (csrc/cin.hip)."""
import torch.nn as nn

from .basemodel import BaseModel
from ..layers import CIN


class xDeepFM(BaseModel):
    """Same arguments as the reference (models/xdeepfm.py:42-45)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 256),
                 cin_layer_size=(256, 128,), cin_split_half=True, cin_activation='relu', l2_reg_linear=0.00001,
                 l2_reg_embedding=0.00001, l2_reg_dnn=0, l2_reg_cin=0, init_std=0.0001, seed=1024, dnn_dropout=0,
                 dnn_activation='relu', dnn_use_bn=False, task='binary', device='cpu', gpus=None):
        super(xDeepFM, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                                      l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                      device=device, gpus=gpus)
        self.dnn_hidden_units = dnn_hidden_units
        self.use_dnn = len(dnn_feature_columns) > 0 and len(dnn_hidden_units) > 0
        if self.use_dnn:
            self._make_tower(self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, dnn_activation, l2_reg_dnn,
                             dnn_dropout, dnn_use_bn, init_std, device)

        self.cin_layer_size = cin_layer_size
        self.use_cin = len(self.cin_layer_size) > 0 and len(dnn_feature_columns) > 0
        if self.use_cin:
            field_num = len(self.embedding_dict)       # counts tables, like the reference (xdeepfm.py:65)
            if cin_split_half == True:  # noqa: E712
                self.featuremap_num = sum(cin_layer_size[:-1]) // 2 + cin_layer_size[-1]
            else:
                self.featuremap_num = sum(cin_layer_size)
            self.cin = CIN(field_num, cin_layer_size, cin_activation, cin_split_half, l2_reg_cin, seed, device=device)
            self.cin_linear = nn.Linear(self.featuremap_num, 1, bias=False).to(device)
            self.add_regularization_weight(filter(lambda x: 'weight' in x[0], self.cin.named_parameters()),
                                           l2=l2_reg_cin)
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        dnn_input, linear_logit, _ = self.fused_inputs(X, want_fm=False, full=True)
        parts = [linear_logit]
        if self.use_cin:
            if plan.emb_dim <= 0:
                raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
            B, nf = X.shape[0], len(plan.deep)
            hooked = any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks
                         for m in (self.cin, self.cin_linear))
            if not hooked and self.cin._stack_ok(nf) and type(self.cin_linear) is nn.Linear:
                # CIN + cin_linear as one autograd node on the gather's row matrix itself: the gradient comes back in
                # that shape, the projection is a wave-per-row dot product forward and folded into the layers'
                # gradient assembly backward (_hip/ops.py CINStackFunction)
                parts.append(self.cin._stack(dnn_input, nf, plan.emb_dim, self.cin_linear.weight))
            else:
                cin_input = dnn_input[:, :plan.emb_width].reshape(B, nf, plan.emb_dim)   # view of the gather's output
                parts.append(self.cin_linear(self.cin(cin_input)))
        if self.use_dnn:
            parts.append(self.tower_logit(dnn_input, plan.width))
        return parts
