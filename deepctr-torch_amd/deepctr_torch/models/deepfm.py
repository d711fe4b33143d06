# -*- coding: utf-8 -*-
"""DeepFM (reference models/deepfm.py:16-86): linear + FM + DNN over shared embeddings.

The forward is two device phases instead of the reference's ~200 ATen launches:
one fused gather (embeddings in DNN-input layout + linear logit + FM term) and the MLP tower."""

from .basemodel import BaseModel
from ..layers import FM


class DeepFM(BaseModel):
    """Same arguments as the reference (models/deepfm.py:38-43)."""
    _fused_step_ok = True
    _gather_step = True     # logit_parts() is [linear, (fm), tower] over ONE fused lookup: _hip/step.py applies

    def __init__(self, linear_feature_columns, dnn_feature_columns, use_fm=True, dnn_hidden_units=(256, 128),
                 l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_dnn=0, init_std=0.0001, seed=1024,
                 dnn_dropout=0, dnn_activation='relu', dnn_use_bn=False, task='binary', device='cpu', gpus=None):
        super(DeepFM, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                                     l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                     device=device, gpus=gpus)
        self.use_fm = use_fm
        self.use_dnn = len(dnn_feature_columns) > 0 and len(dnn_hidden_units) > 0
        if use_fm:
            self.fm = FM()
        if self.use_dnn:
            self._make_tower(self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, dnn_activation, l2_reg_dnn,
                             dnn_dropout, dnn_use_bn, init_std, device)
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        want_fm = self.use_fm and len(plan.deep) > 0
        dnn_input, logit, fm_logit = self.fused_inputs(X, want_fm=want_fm, full=self.use_dnn)
        parts = [logit]
        if want_fm:
            parts.append(fm_logit)
        if self.use_dnn:
            parts.append(self.tower_logit(dnn_input, plan.width))
        return parts
