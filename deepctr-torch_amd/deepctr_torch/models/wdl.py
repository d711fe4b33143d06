# -*- coding: utf-8 -*-
"""WDL -- Wide & Deep (reference models/wdl.py:15-75): the linear ("wide") logit + the DNN tower over the embeddings.
One fused gather (embeddings in DNN-input layout + the wide logit) and the MFMA tower; takes the fused train step."""

from .basemodel import BaseModel


class WDL(BaseModel):
    """Same arguments as the reference (models/wdl.py:36-41)."""
    _fused_step_ok = True
    _gather_step = True     # logit_parts() is [linear, (fm), tower] over ONE fused lookup: _hip/step.py applies

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 128), l2_reg_linear=1e-5,
                 l2_reg_embedding=1e-5, l2_reg_dnn=0, init_std=0.0001, seed=1024, dnn_dropout=0, dnn_activation='relu',
                 dnn_use_bn=False, task='binary', device='cpu', gpus=None):
        super(WDL, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                                  l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                  device=device, gpus=gpus)
        self.use_dnn = len(dnn_feature_columns) > 0 and len(dnn_hidden_units) > 0
        if self.use_dnn:
            self._make_tower(self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, dnn_activation, l2_reg_dnn,
                             dnn_dropout, dnn_use_bn, init_std, device)
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        dnn_input, logit, _ = self.fused_inputs(X, want_fm=False, full=self.use_dnn)
        parts = [logit]
        if self.use_dnn:
            parts.append(self.tower_logit(dnn_input, plan.width))
        return parts
