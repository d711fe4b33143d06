# -*- coding: utf-8 -*-
"""AFM (reference models/afm.py:16-80): linear + attentional pooling of the pairwise embedding products.

Forward = the fused gather (embeddings + linear logit, and the FM term when ``use_attention=False``) and ONE kernel
for the whole AFMLayer (csrc/afm.hip)."""
from .basemodel import BaseModel
from ..layers import AFMLayer, FM


class AFM(BaseModel):
    """Same arguments as the reference (models/afm.py:35-37)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, use_attention=True, attention_factor=8,
                 l2_reg_linear=1e-5, l2_reg_embedding=1e-5, l2_reg_att=1e-5, afm_dropout=0, init_std=0.0001, seed=1024,
                 task='binary', device='cpu', gpus=None):
        super(AFM, self).__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                                  l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                                  device=device, gpus=gpus)
        from ..inputs import DenseFeat
        if any(isinstance(c, DenseFeat) for c in dnn_feature_columns):     # reference: support_dense=False, afm.py:62-63
            raise ValueError("DenseFeat is not supported in dnn_feature_columns")
        self.use_attention = use_attention
        if use_attention:
            self.fm = AFMLayer(self.embedding_size, attention_factor, l2_reg_att, afm_dropout, seed, device)
            self.add_regularization_weight(self.fm.attention_W, l2=l2_reg_att)
        else:
            self.fm = FM()
        self.to(device)

    def logit_parts(self, X):
        plan = self.model_plan()
        has_emb = len(plan.deep) > 0
        gathered, logit, fm_logit = self.fused_inputs(X, want_fm=(has_emb and not self.use_attention))
        parts = [logit]
        if has_emb:
            if self.use_attention:
                if plan.emb_dim <= 0:
                    raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
                emb = gathered[:, :plan.emb_width].reshape(X.shape[0], len(plan.deep), plan.emb_dim)
                parts.append(self.fm(emb))
            else:
                parts.append(fm_logit)
        return parts
