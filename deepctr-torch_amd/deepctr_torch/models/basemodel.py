# -*- coding: utf-8 -*-
"""``Linear`` + ``BaseModel``: the Keras-style surface of the reference (models/basemodel.py:34-527)
over the MI355X hot path.

Kept verbatim: constructor / ``compile`` / ``fit`` / ``evaluate`` / ``predict`` signatures, attribute
names touched by callbacks and tests (``embedding_dict``, ``linear_model``, ``feature_index``,
``history``, ``stop_training``, ``regularization_weight`` ...) and every ``state_dict`` key.

Re-designed underneath:
  * lookups, pooling, the wide logit and FM run as ONE gfx950 kernel per direction (``_hip.ops.embed``);
  * the embedding backward is an O(batch) scatter; with ``compile('sgd'|'adagrad')`` and
    ``l2_reg_embedding = l2_reg_linear = 0`` the optimizer update of the touched rows is fused into it
    (identical result to the reference's dense update, because rows with zero gradient do not move under
    those optimizers); every other optimizer / regulariser gets the exact dense gradient in ``param.grad``;
  * ``fit`` keeps the whole dataset resident in HBM (a 45 M-row Criteo day is 7 GB of 288), batches are
    index-selected on the device, the loss is accumulated on the device and read back once per epoch.
"""
from __future__ import print_function

import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from sklearn.metrics import accuracy_score, log_loss, mean_squared_error, roc_auc_score

try:  # tqdm is optional: only the verbose=1 progress bar needs it
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None

from .. import callbacks as _cb
from .._hip import dense as _dense
from .._hip import mlp as _mlp
from .._hip import ops as _ops
from .._hip import step as _step
from .._hip.plan import EmbeddingPlan
from ..inputs import SparseFeat, VarLenSparseFeat, build_input_features, create_embedding_matrix, split_columns
from ..layers import PredictionLayer
from ..layers.utils import slice_arrays


class Linear(nn.Module):
    """First-order ("wide") logit: sum of 1-dim embeddings (+ pooled VarLen) + dense . weight
    (reference basemodel.py:34-92).  Computed by the gather kernel's wide path."""

    def __init__(self, feature_columns, feature_index, init_std=0.0001, device='cpu'):
        super(Linear, self).__init__()
        self.feature_index = feature_index
        self.device = device
        self.sparse_feature_columns, self.varlen_sparse_feature_columns, self.dense_feature_columns = \
            split_columns(feature_columns)
        self._columns = list(feature_columns)
        self.embedding_dict = create_embedding_matrix(feature_columns, init_std, linear=True, sparse=False,
                                                      device=device)
        # the reference initialises these tables a second time (basemodel.py:55-56); doing the same keeps
        # same-seed initial weights identical to the reference's
        # (drawn from the CPU generator whatever the device -- see the note on self.weight below)
        for tensor in self.embedding_dict.values():
            w = torch.empty(tensor.weight.shape, dtype=tensor.weight.dtype)
            nn.init.normal_(w, mean=0, std=init_std)
            with torch.no_grad():
                tensor.weight.copy_(w)
        if len(self.dense_feature_columns) > 0:
            # The reference draws this one on the target device (basemodel.py:58-61): on a GPU that is the CUDA
            # generator, whose stream no ROCm build can reproduce.  Drawn from the CPU generator instead, the same
            # seed gives the weights (and every later draw) of the reference constructed on its CPU device, whatever
            # device this model lives on.
            w = torch.Tensor(sum(fc.dimension for fc in self.dense_feature_columns), 1)
            torch.nn.init.normal_(w, mean=0, std=init_std)
            self.weight = nn.Parameter(w.to(device))
        self._plan = None

    def plan(self):
        if self._plan is None:
            self._plan = EmbeddingPlan(self.feature_index, wide_columns=self._columns,
                                       wide_tables=self.embedding_dict,
                                       wide_dense_weight=getattr(self, "weight", None))
            owner = getattr(self.embedding_dict, "_dctr_owner_plan", None)
            if owner is not None:
                self._plan.share_update_with(owner)
        return self._plan

    def forward(self, X, sparse_feat_refine_weight=None):
        if sparse_feat_refine_weight is not None:
            return self._forward_refined(X, sparse_feat_refine_weight)
        plan = self.plan()
        if not plan.has_wide:
            return torch.zeros([X.shape[0], 1], device=X.device)
        _, wide, _ = _ops.embed(plan, X)
        return wide.unsqueeze(1)


    def _forward_refined(self, X, refine):
        """w_{x,i} = m_{x,i} * w_i (IFM / DIFM; reference basemodel.py:80-91).  The per-field first-order weights come
        out of ONE fused gather as a [B, n] matrix (the tables seen as 1-dim deep fields; VarLen fields pooled), then the
        reference's own operations in its own order: cat * m_x, sum over the fields, + dense . weight.  Gradients:
        autograd hands the gather d logit * m_x per field -- the fused table update takes it from there -- and m_x
        its d logit * w."""
        cols = list(self.sparse_feature_columns) + list(self.varlen_sparse_feature_columns)
        linear_logit = torch.zeros([X.shape[0], 1], device=X.device)
        if cols:
            embs = _ops.gather_columns(X, self.embedding_dict, self.feature_index, cols)      # each [B, 1, 1]
            cat = torch.cat(embs, dim=-1) * refine.unsqueeze(1)
            linear_logit = linear_logit + torch.sum(cat, dim=-1, keepdim=False)
        if len(self.dense_feature_columns) > 0:
            dense = torch.cat([X[:, self.feature_index[fc.name][0]:self.feature_index[fc.name][1]]
                               for fc in self.dense_feature_columns], dim=-1)
            linear_logit = linear_logit + dense.matmul(self.weight)
        return linear_logit


class BaseModel(nn.Module):
    def __init__(self, linear_feature_columns, dnn_feature_columns, l2_reg_linear=1e-5, l2_reg_embedding=1e-5,
                 init_std=0.0001, seed=1024, task='binary', device='cpu', gpus=None):
        super(BaseModel, self).__init__()
        torch.manual_seed(seed)
        self.dnn_feature_columns = dnn_feature_columns
        self.reg_loss = torch.zeros((1,), device=device)
        self.aux_loss = torch.zeros((1,), device=device)
        self.device = device
        self.gpus = gpus
        if gpus and str(self.gpus[0]) not in self.device:
            raise ValueError("`gpus[0]` should be the same gpu with `device`")

        self.feature_index = build_input_features(linear_feature_columns + dnn_feature_columns)
        self._linear_feature_columns = list(linear_feature_columns)
        self.embedding_dict = create_embedding_matrix(dnn_feature_columns, init_std, sparse=False, device=device)
        self.linear_model = Linear(linear_feature_columns, self.feature_index, device=device)

        self.regularization_weight = []
        self._n_embedding_reg_groups = 2
        self.add_regularization_weight(self.embedding_dict.parameters(), l2=l2_reg_embedding)
        self.add_regularization_weight(self.linear_model.parameters(), l2=l2_reg_linear)

        self.out = PredictionLayer(task, )
        self.to(device)

        self._is_graph_network = True   # attributes Keras-style callbacks look for
        self._ckpt_saved_epoch = False
        self.history = _cb.History()
        self.stop_training = False
        self._plan = None
        self._fit_graph = None    # hipGraph of the train step used by fit() for full-size batches
        self._defer_tower = False
        self._grad_sink = None    # dense.DenseSlab while a fused train step runs
        self._fused = None        # cached state of the fused train step (see _fused_step_state)

    def __getstate__(self):
        # captured hipGraphs hold raw device handles: never pickled, re-captured on demand
        self._flush_lazy()       # the pickled tables are the reference's tables
        d = dict(self.__dict__)
        d["_fit_graph"] = None
        if d.get("_fused"):       # the step engine holds ctypes descriptors and device buffers: rebuilt on demand
            d["_fused"] = {k: v for k, v in d["_fused"].items() if k != "engine"}
        return d

    def _flush_lazy(self):
        """Tables on the exact lazy update (csrc/lazy.hip) hold rows that are behind the optimizer's step count:
        bring them up to date before anything outside the train step looks at them."""
        plan = self.__dict__.get("_plan")
        if plan is not None and plan._lazy is not None:
            plan._lazy.flush()
        fused = self.__dict__.get("_fused")
        if fused and fused.get("slab") is not None and not (
                fused["slab"].flat.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            fused["slab"].sync_optimizer_state()      # Adam's per-parameter `step` entries

    def state_dict(self, *args, **kwargs):
        """The reference's keys / shapes / values -- as COPIES for every entry that is a view of a larger storage (all
        interleaved Adagrad tables, all dense parameters of a fused-step model): unlike ``nn.Module.state_dict()`` those
        entries do not alias the parameters (an in-place edit of one is not seen by the model: use ``load_state_dict``),
        and a call allocates a transient copy of every such table on the device (``keep_vars=True`` returns the aliasing
        views).  A table the fused Adagrad update seated in an interleaved slab
        (_hip/layout.py: row and optimizer state share a 128-byte line) is a strided VIEW; torch.save would serialise the
        whole underlying storage -- 2-4x the table's bytes, optimizer accumulators included (round-2 advisor finding).
        The dense parameters of a fused-step model are views of ONE flat slab likewise.  Every entry that is a view of
        something bigger is handed out as a contiguous copy: a weights-only checkpoint is byte-for-byte the reference's."""
        self._flush_lazy()
        sd = super(BaseModel, self).state_dict(*args, **kwargs)
        if kwargs.get("keep_vars", False):
            return sd
        for k, v in list(sd.items()):
            # a view of something bigger (a table in its slab, a dense parameter in the flat slab of the fused step)
            if torch.is_tensor(v) and v.untyped_storage().nbytes() > v.numel() * v.element_size():
                sd[k] = v.detach().clone(memory_format=torch.contiguous_format)
        return sd

    def load_state_dict(self, *args, **kwargs):
        self._flush_lazy()       # every stamp == the step counter: the loaded rows are current by definition
        return super(BaseModel, self).load_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------------------------------------
    # hot path entry points
    # ------------------------------------------------------------------------------------------------
    def model_plan(self):
        """The model-wide compiled schema: deep side = ``dnn_feature_columns`` over ``embedding_dict``,
        wide side = ``linear_feature_columns`` over ``linear_model.embedding_dict``."""
        if self._plan is None:
            lm = self.linear_model
            self._plan = EmbeddingPlan(self.feature_index, deep_columns=self.dnn_feature_columns,
                                       deep_tables=self.embedding_dict,
                                       wide_columns=self._linear_feature_columns, wide_tables=lm.embedding_dict,
                                       wide_dense_weight=getattr(lm, "weight", None))
            object.__setattr__(self.embedding_dict, "_dctr_owner_plan", self._plan)
            object.__setattr__(lm.embedding_dict, "_dctr_owner_plan", self._plan)
            if lm._plan is not None:
                lm._plan.share_update_with(self._plan)
            self._apply_update_mode()
        elif self._plan.update[0] == "lazy" and self._plan._lazy is None:
            self._apply_update_mode()        # an unpickled model: the lazy state is rebuilt from the optimizer
        return self._plan

    def fused_inputs(self, X, want_fm=False, full=False):
        """One kernel launch -> (dnn_input ``[B, sum(D)+n_dense]``, linear logit ``[B, 1]``, FM ``[B, 1]``).

        ``dnn_input`` is exactly ``combined_dnn_input(*input_from_feature_columns(...))`` of the reference
        (its first ``sum(D)`` columns viewed ``[B, F, D]`` are the ``torch.cat(sparse_embedding_list, 1)``
        every interaction layer consumes), the logit is ``self.linear_model(X)``, and FM is
        ``FM()(that view)`` (deepfm.py:69-82)."""
        plan = self.model_plan()
        out, wide, fm = _ops.embed(plan, X, want_fm=want_fm, full=full)
        return out, wide.unsqueeze(1), fm.unsqueeze(1)

    def tower_logit(self, x, K=None):
        """``self.dnn_linear(self.dnn(x[:, :K]))`` on the MFMA tower kernels (csrc/mlp.hip) when the tower is
        relu / linear without BatchNorm and dropout is inactive, else through the modules."""
        if self._defer_tower and self._grad_sink is not None:
            # fused train step: the tower runs later, together with the head and its own backward
            return _mlp.PendingTower(self.dnn, self.dnn_linear, x, x.shape[1] if K is None else K, self._grad_sink)
        return _mlp.tower(self.dnn, self.dnn_linear, x, K, sink=self._grad_sink)

    def fused_loss(self, xb, yb, slab):
        """(loss, y_pred) of the fused train step: ``logit_parts`` with the tower deferred, then tower + head + BCE
        (+ the tower's whole backward) as one op when the tower is the LAST summand and at most two parts precede
        it (every model of this package), else tower and head as two ops."""
        self._defer_tower = os.environ.get("DCTR_FUSED_HEAD", "1") != "0"
        try:
            parts = self.logit_parts(xb)
        finally:
            self._defer_tower = False
        pend = parts[-1] if isinstance(parts[-1], _mlp.PendingTower) else None
        if any(isinstance(p, _mlp.PendingTower) for p in parts[:-1]):
            raise RuntimeError("logit_parts() must list the tower's logit last")
        if pend is not None and _mlp.fusable_head(pend, parts[:-1]):
            return _mlp.tower_head(pend, parts[:-1], self.out.bias, yb)
        if pend is not None:
            parts = list(parts[:-1]) + [_mlp.tower(pend.dnn, pend.dnn_linear, pend.x, pend.K, sink=pend.sink)]
        return _mlp.bce_head(parts, self.out.bias, yb, unit=True, g_bias_sink=slab.grad_of(self.out.bias))

    def tower_hidden(self, x, K=None):
        """``self.dnn(x[:, :K])`` (no projection) on the same kernels."""
        return _mlp.tower(self.dnn, None, x, K)

    def logit_parts(self, X):
        """The summands of the final logit, in the reference's order of addition.  Models override this;
        ``forward`` is ``self.out(sum(parts))``."""
        raise NotImplementedError

    def forward(self, X):
        parts = self.logit_parts(X)
        logit = parts[0]
        for p in parts[1:]:
            logit = logit + p
        return self.out(logit)

    def input_from_feature_columns(self, X, feature_columns, embedding_dict, support_dense=True):
        """Reference-shaped accessor (basemodel.py:354-380): list of ``[B, 1, D]`` embeddings (fixed-length
        features, then pooled VarLen features) and list of dense ``[B, dim]`` slices of X."""
        sparse_cols, varlen_cols, dense_cols = split_columns(feature_columns)
        if not support_dense and len(dense_cols) > 0:
            raise ValueError("DenseFeat is not supported in dnn_feature_columns")
        emb_cols = sparse_cols + varlen_cols
        emb_list = _ops.gather_columns(X, embedding_dict, self.feature_index, emb_cols, pooled=True) \
            if emb_cols else []
        dense_value_list = [X[:, self.feature_index[fc.name][0]:self.feature_index[fc.name][1]] for fc in dense_cols]
        return emb_list, dense_value_list

    def _make_tower(self, in_features, hidden_units, activation='relu', l2_reg_dnn=0, dropout=0, use_bn=False,
                    init_std=0.0001, device='cpu', head_in=None, l2_head=None, head_first=False):
        """``self.dnn`` (+ ``self.dnn_linear``, the bias-free 1-unit projection every model of the family puts on top of
        it) with their regularisation groups -- the block each reference model spells out in its constructor (e.g.
        deepfm.py:55-63, dcn.py:53-70).  ``head_in``: input width of ``dnn_linear`` when it reads more than the tower's
        last layer (DCN / AutoInt stack other features beside it); ``l2_head``: its L2 strength (default: l2_reg_dnn),
        ``False`` leaves it unregularised; ``head_first``: construct ``dnn_linear`` before ``dnn`` (autoint.py:63-69) --
        the order in which the two consume the random generator decides the initial weights a seed produces."""
        from ..layers import DNN

        def head():
            return nn.Linear(hidden_units[-1] if head_in is None else head_in, 1, bias=False).to(device)
        if head_first:
            self.dnn_linear = head()
        self.dnn = DNN(in_features, hidden_units, activation=activation, l2_reg=l2_reg_dnn, dropout_rate=dropout,
                       use_bn=use_bn, init_std=init_std, device=device)
        if not head_first:
            self.dnn_linear = head()
        self.add_regularization_weight(
            [kv for kv in self.dnn.named_parameters() if 'weight' in kv[0] and 'bn' not in kv[0]], l2=l2_reg_dnn)
        if l2_head is not False:
            self.add_regularization_weight(self.dnn_linear.weight, l2=l2_reg_dnn if l2_head is None else l2_head)

    def compute_input_dim(self, feature_columns, include_sparse=True, include_dense=True, feature_group=False):
        dense_cols = split_columns(feature_columns)[2]
        emb_cols = [c for c in feature_columns if isinstance(c, (SparseFeat, VarLenSparseFeat))] \
            if len(feature_columns) else []
        dense_dim = sum(c.dimension for c in dense_cols)
        sparse_dim = len(emb_cols) if feature_group else sum(c.embedding_dim for c in emb_cols)
        return (sparse_dim if include_sparse else 0) + (dense_dim if include_dense else 0)

    @property
    def embedding_size(self):
        dims = set(c.embedding_dim for c in self.dnn_feature_columns
                   if isinstance(c, (SparseFeat, VarLenSparseFeat))) if len(self.dnn_feature_columns) else set()
        if len(dims) > 1:
            raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
        return list(dims)[0]

    # ------------------------------------------------------------------------------------------------
    # regularisation / auxiliary loss (reference basemodel.py:402-431)
    # ------------------------------------------------------------------------------------------------
    def add_regularization_weight(self, weight_list, l1=0.0, l2=0.0):
        if isinstance(weight_list, torch.nn.parameter.Parameter):
            weight_list = [weight_list]
        else:  # generators / filters must become lists so the model stays picklable
            weight_list = list(weight_list)
        self.regularization_weight.append((weight_list, l1, l2))
        if self.__dict__.get("optim") is not None:
            # added after compile(): the reference evaluates its regularisers at every step, so the new term counts from
            # the next step on -- re-derive which update path is still exact (lazy / in-kernel / fused need L2-only terms)
            self._rederive_update_paths()

    def get_regularization_loss(self):
        total = torch.zeros((1,), device=self.device)
        # tables on the exact lazy update (csrc/lazy.hip): their L2 GRADIENT is applied by the kernels, their term of
        # the logged loss comes from LazyState.reg_value -- never a dense O(vocabulary) autograd node
        lazy = self._plan.lazy if (self._plan is not None and self._plan.update[0] == "lazy") else None
        lazy_ids = set(id(p) for p in self._plan.table_params) if lazy is not None else ()
        if lazy is not None:
            rv = lazy.reg_value(self.device)
            if rv is not None:
                total = total + rv
        for weight_list, l1, l2 in self.regularization_weight:
            if not (l1 > 0 or l2 > 0):
                continue
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w  # named_parameters() yields (name, tensor)
                if id(p) in lazy_ids:
                    continue
                if l1 > 0:
                    total = total + torch.sum(l1 * torch.abs(p))
                if l2 > 0:
                    total = total + torch.sum(l2 * torch.square(p))
        return total

    def add_auxiliary_loss(self, aux_loss, alpha):
        self.aux_loss = aux_loss * alpha
        self._aux_default = False

    def _aux_is_default(self):
        return getattr(self, "_aux_default", True)

    # ------------------------------------------------------------------------------------------------
    # compile (reference basemodel.py:433-516)
    # ------------------------------------------------------------------------------------------------
    def compile(self, optimizer, loss=None, metrics=None):
        self.metrics_names = ["loss"]
        self.optim = self._get_optim(optimizer)
        self.loss_func = self._get_loss_func(loss)
        self.metrics = self._get_metrics(metrics)
        self._apply_update_mode()
        self._hyper_sig = self._hyper_raw = None
        self._sync_optimizer_hyper()          # records the values the paths were derived from

    _HYPER_KEYS = ("lr", "eps", "betas", "weight_decay", "momentum", "dampening", "nesterov", "lr_decay", "alpha",
                   "centered", "amsgrad", "maximize", "initial_accumulator_value")

    def _optim_signature(self):
        """The optimizer's hyper-parameters as a comparable value.  The O(batch) paths bake lr / eps / betas into
        kernel arguments (and into captured hipGraphs); the reference reads ``param_groups`` at every step, so a
        learning-rate schedule stepping ``model.optim`` between epochs must be honoured here too."""
        opt = self.__dict__.get("optim")
        if opt is None:
            return None
        sig = []
        for grp in opt.param_groups:
            row = []
            for k in self._HYPER_KEYS:
                v = grp.get(k)
                if torch.is_tensor(v):
                    v = float(v) if v.numel() == 1 and not v.is_cuda else id(v)
                row.append(tuple(v) if isinstance(v, (list, tuple)) else v)
            sig.append(tuple(row))
        return tuple(sig)

    def _sync_optimizer_hyper(self, full=True):
        """Re-derive the update paths when the optimizer's hyper-parameters (or, with ``full``, the set of frozen
        tables) changed since they were last looked at; lazily replayed rows are first flushed with the OLD values.
        The common case -- nothing changed -- is one list comparison (~1 us): this runs at every step of fit()."""
        opt = self.__dict__.get("optim")
        if opt is None:
            return
        raw = [grp.get(k) for grp in opt.param_groups for k in self._HYPER_KEYS]
        raw.append(id(opt))
        raw.append(id(self.__dict__.get("loss_func")))          # the fused step bakes in BCE
        raw.append(len(self.regularization_weight))            # (appended to directly, not via add_regularization_weight)
        if full and self._plan is not None:
            raw.append(tuple(p.requires_grad for p in self._plan.table_params))
        else:
            raw.append(None)
        cached = self.__dict__.get("_hyper_raw")
        try:
            same = cached is not None and raw[:-1] == cached[:-1] and (raw[-1] is None or raw[-1] == cached[-1])
        except Exception:           # tensor-valued hyper-parameters do not compare with ==
            same = False
        if same:
            return
        if raw[-1] is None and cached is not None:
            raw[-1] = cached[-1]
        self._hyper_raw = raw
        sig = (self._optim_signature(), raw[-4], raw[-3], raw[-2], raw[-1])   # (+ optimizer object, loss, #regularisers, frozen tables)
        if sig != self.__dict__.get("_hyper_sig"):
            prev = self.__dict__.get("_hyper_sig")
            first = prev is None
            # (recorded before the plan existed -- compile() ahead of the first lookup: the frozen-table entry was None then.
            # Nothing was derived from it; filling it in is no change.  Re-deriving here dropped the fused-step state, and
            # with it the dense slab a ShardedTrainer had already adopted: fit() under torchrun)
            filled_in = (not first) and prev[-1] is None and sig[:-1] == prev[:-1]
            self._hyper_sig = sig
            if not first and not filled_in:
                self._rederive_update_paths()

    def _rederive_update_paths(self):
        """Something the O(batch) paths were derived from changed (hyper-parameters, the set of regularisers): bring
        every lazily replayed row and the optimizer's ``step`` entries up to date under the OLD settings, drop the
        fused-step state and any captured step, choose the paths again."""
        self._flush_lazy()
        self._fused = None
        self._fit_graph = None
        self._apply_update_mode()

    def _get_optim(self, optimizer):
        if not isinstance(optimizer, str):
            return optimizer
        if optimizer == "sgd":
            return torch.optim.SGD(self.parameters(), lr=0.01)
        if optimizer == "adam":
            return torch.optim.Adam(self.parameters())  # 0.001
        if optimizer == "adagrad":
            return torch.optim.Adagrad(self.parameters())  # 0.01
        if optimizer == "rmsprop":
            return torch.optim.RMSprop(self.parameters())
        raise NotImplementedError

    def _get_loss_func(self, loss):
        if isinstance(loss, str):
            return self._get_loss_func_single(loss)
        if isinstance(loss, list):
            return [self._get_loss_func_single(l) for l in loss]
        return loss

    def _get_loss_func_single(self, loss):
        table = {"binary_crossentropy": F.binary_cross_entropy, "mse": F.mse_loss, "mae": F.l1_loss}
        if loss not in table:
            raise NotImplementedError
        return table[loss]

    def _log_loss(self, y_true, y_pred, eps=1e-7, normalize=True, sample_weight=None, labels=None):
        y_pred = np.clip(np.asarray(y_pred, dtype=np.float64), eps, 1 - eps)
        return log_loss(y_true, y_pred, normalize=normalize, sample_weight=sample_weight, labels=labels)

    @staticmethod
    def _accuracy_score(y_true, y_pred):
        return accuracy_score(y_true, np.where(y_pred > 0.5, 1, 0))

    def _get_metrics(self, metrics, set_eps=False):
        chosen = {}
        for metric in (metrics or []):
            if metric in ("binary_crossentropy", "logloss"):
                chosen[metric] = self._log_loss if set_eps else log_loss
            if metric == "auc":
                chosen[metric] = roc_auc_score
            if metric == "mse":
                chosen[metric] = mean_squared_error
            if metric in ("accuracy", "acc"):
                chosen[metric] = self._accuracy_score
            self.metrics_names.append(metric)
        return chosen

    def _in_multi_worker_mode(self):
        return None

    # ------------------------------------------------------------------------------------------------
    # which embedding update runs inside backward
    # ------------------------------------------------------------------------------------------------
    def _embedding_reg_active(self):
        return any((l1 > 0 or l2 > 0) for (_, l1, l2) in self.regularization_weight[:self._n_embedding_reg_groups])

    def _sparse_update_mode(self):
        """("sgd", lr) / ("adagrad", lr, eps) when the fused O(batch) update is EXACTLY the reference's dense
        update, otherwise ("dense",)."""
        opt = getattr(self, "optim", None)
        if opt is None or self._plan is None or os.environ.get("DCTR_SPARSE_UPDATE", "1") == "0":
            return ("dense",), {}
        tables = self._plan.table_params
        if not tables or self._embedding_reg_active() or not all(p.requires_grad for p in tables):
            return ("dense",), {}         # (a frozen table: the in-kernel optimizers would move it)
        group_of = {}
        for grp in opt.param_groups:
            for p in grp["params"]:
                group_of[id(p)] = grp
        groups = [group_of.get(id(p)) for p in tables]
        if any(g is None for g in groups):
            return ("dense",), {}
        g0 = groups[0]

        def same(key):
            return all(g.get(key) == g0.get(key) for g in groups)

        if type(opt) is torch.optim.SGD:
            ok = same("lr") and all(g.get("momentum", 0) == 0 and g.get("weight_decay", 0) == 0 and
                                    not g.get("nesterov", False) and not g.get("maximize", False) for g in groups)
            if ok:
                return ("sgd", float(g0["lr"])), {}
        if type(opt) is torch.optim.Adagrad:
            ok = same("lr") and same("eps") and all(g.get("lr_decay", 0) == 0 and g.get("weight_decay", 0) == 0 and
                                                    not g.get("maximize", False) for g in groups)
            if ok and all("sum" in opt.state.get(p, {}) for p in tables):
                return ("adagrad", float(g0["lr"]), float(g0["eps"])), {p: opt.state[p]["sum"] for p in tables}
        return ("dense",), {}

    def _lazy_update_mode(self):
        """("lazy", kind) + the LazyState arguments when the tables can take the EXACT lazy form of the reference's
        dense regularised / Adam / RMSprop update (csrc/lazy.hip): fixed-length fields over distinct tables, a plain SGD /
        Adagrad / Adam / RMSprop over all tables, L2-only regularisation of the tables.  None otherwise."""
        opt = getattr(self, "optim", None)
        plan = self._plan
        if opt is None or plan is None or os.environ.get("DCTR_LAZY_UPDATE", "1") == "0" or \
                os.environ.get("DCTR_SPARSE_UPDATE", "1") == "0" or getattr(self, "_no_lazy_update", False):
            return None
        tables = plan.table_params
        if not tables or not plan.simple_units or plan.max_dim > 64 * (4 if plan.vec == 4 else 1) or \
                not all(p.requires_grad for p in tables):
            return None
        l2 = {}
        tids = set(id(p) for p in tables)
        for weight_list, l1, l2v in self.regularization_weight:
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                if id(p) in tids:
                    if l1 > 0:
                        return None
                    l2[p] = l2.get(p, 0.0) + float(l2v)
        group_of = {}
        for grp in opt.param_groups:
            for p in grp["params"]:
                group_of[id(p)] = grp
        groups = [group_of.get(id(p)) for p in tables]
        if any(g is None for g in groups):
            return None
        g0 = groups[0]

        def same(key):
            return all(g.get(key) == g0.get(key) for g in groups)

        plain = all(g.get("weight_decay", 0) == 0 and not g.get("maximize", False) for g in groups)
        if not plain or not same("lr"):
            return None
        if type(opt) is torch.optim.SGD:
            if all(g.get("momentum", 0) == 0 and not g.get("nesterov", False) for g in groups):
                return dict(kind="sgd", lr=g0["lr"], eps=0.0, beta1=0.0, beta2=0.0, l2=l2, s1={}, s2={})
        if type(opt) is torch.optim.Adagrad:
            if same("eps") and all(g.get("lr_decay", 0) == 0 for g in groups) and \
                    all("sum" in opt.state.get(p, {}) for p in tables):
                return dict(kind="adagrad", lr=g0["lr"], eps=g0["eps"], beta1=0.0, beta2=0.0, l2=l2,
                            s1={p: opt.state[p]["sum"] for p in tables}, s2={})
        if type(opt) is torch.optim.RMSprop:
            if same("eps") and same("alpha") and all(g.get("momentum", 0) == 0 and not g.get("centered", False) and
                                                       not g.get("capturable", False) for g in groups):
                for p in tables:      # torch creates the state at the first step(); the kernels need it now
                    st = opt.state[p]
                    if "square_avg" not in st:
                        st["step"] = torch.tensor(0.0, dtype=torch.float32)
                        st["square_avg"] = torch.zeros_like(p.data)
                # (beta1 carries 1 - alpha, rounded from double like the scalar torch hands its kernels)
                return dict(kind="rmsprop", lr=g0["lr"], eps=g0["eps"], beta1=1 - g0["alpha"], beta2=g0["alpha"], l2=l2,
                            s1={p: opt.state[p]["square_avg"] for p in tables}, s2={})
        if type(opt) is torch.optim.Adam:
            if same("eps") and same("betas") and all(not g.get("amsgrad", False) and not g.get("capturable", False) and
                                                       not g.get("fused", False) for g in groups):
                for p in tables:      # torch creates Adam's state at the first step(); the kernels need it now
                    st = opt.state[p]
                    if "exp_avg" not in st:
                        st["step"] = torch.tensor(0.0, dtype=torch.float32)
                        st["exp_avg"] = torch.zeros_like(p.data)
                        st["exp_avg_sq"] = torch.zeros_like(p.data)
                return dict(kind="adam", lr=g0["lr"], eps=g0["eps"], beta1=g0["betas"][0], beta2=g0["betas"][1], l2=l2,
                            s1={p: opt.state[p]["exp_avg"] for p in tables},
                            s2={p: opt.state[p]["exp_avg_sq"] for p in tables})
        return None

    def _apply_update_mode(self):
        if self._plan is None:
            return
        mode, state = self._sparse_update_mode()
        lazy = self._lazy_update_mode() if mode[0] == "dense" else None
        if lazy is not None:
            from .._hip.plan import LazyState
            new = LazyState(self._plan, optimizer=self.optim, **lazy)
            old = self._plan._lazy
            if old is not None and old.signature() == new.signature() and old.optimizer is self.optim:
                old.s1, old.s2 = new.s1, new.s2       # same schedule: keep the stamps and the step counter
                old._key = None
            else:
                if old is not None:
                    old.flush()
                self._plan._lazy = new
            self._plan.set_state({})
            self._plan.ensure_gacc()
            self._plan.update = ("lazy", lazy["kind"])
            return
        if self._plan._lazy is not None:
            self._plan._lazy.flush()
            self._plan._lazy = None
        if mode[0] == "adagrad" and self._plan.unit_path:
            # what the update kernel touches together lives together: a row shares its 128-byte line with its
            # Adagrad state (weights and optimizer state become strided views; _hip/layout.py)
            from .._hip.layout import apply_layout
            state = apply_layout(self._plan, self.optim)
            self._contiguous_optimizer_state_dict()
        self._plan.set_state(state)
        if mode[0] == "adagrad" or (mode[0] == "sgd" and self._plan.has_maxpool):
            self._plan.ensure_gacc()   # two-pass updates: allocate the slabs now (outside any graph capture)
        self._plan.update = mode       # ("dense",) allocates its slabs lazily at the first backward

    def _contiguous_optimizer_state_dict(self):
        """``optimizer.state_dict()`` hands out the Adagrad accumulators that live in an interleaved slab as contiguous
        copies too (same reason as ``state_dict``: a saved optimizer must not drag the tables along)."""
        opt = self.optim
        if getattr(opt, "_dctr_sd_hook", False) or not hasattr(opt, "register_state_dict_post_hook"):
            return

        def hook(optimizer, sd):
            for st in sd.get("state", {}).values():
                for key, v in list(st.items()):
                    if torch.is_tensor(v) and v.untyped_storage().nbytes() > v.numel() * v.element_size():
                        st[key] = v.detach().clone(memory_format=torch.contiguous_format)
            return sd
        opt.register_state_dict_post_hook(hook)
        opt._dctr_sd_hook = True

    # ------------------------------------------------------------------------------------------------
    # data plumbing shared by fit / evaluate / predict
    # ------------------------------------------------------------------------------------------------
    def _as_matrix(self, x):
        """dict / list of per-feature arrays -> one float32 ``[N, sum(widths)]`` matrix on ``self.device``
        (reference basemodel.py:155-156,191-198: np.concatenate in ``feature_index`` order)."""
        if torch.is_tensor(x) and x.dim() == 2:
            # (beyond the reference: a dataset that already is one [N, sum(widths)] matrix -- on the device it is used in
            # place, nothing is concatenated or uploaded again by every fit / predict call)
            need = max([hi for (_, hi) in self.feature_index.values()] + [0])
            if x.shape[1] < need:
                raise ValueError("the input matrix has %d columns, the feature columns need %d" % (x.shape[1], need))
            return x.to(self.device).float()
        if isinstance(x, dict):
            x = [x[feature] for feature in self.feature_index]
        x = list(x)
        for i in range(len(x)):
            if len(x[i].shape) == 1:
                x[i] = np.expand_dims(x[i], axis=1)
        return torch.from_numpy(np.concatenate(x, axis=-1)).to(self.device).float()

    # ------------------------------------------------------------------------------------------------
    # fused train step: tower + head + dense optimizer on the C-ABI kernels (launch count)
    # ------------------------------------------------------------------------------------------------
    def _dense_update_mode(self, params):
        """("sgd", lr) / ("adagrad", lr, eps) when one fused pass over the dense slab is EXACTLY what the
        compiled torch optimizer would do to ``params``, else None."""
        opt = getattr(self, "optim", None)
        if opt is None or not params:
            return None
        group_of = {}
        for grp in opt.param_groups:
            for p in grp["params"]:
                group_of[id(p)] = grp
        groups = [group_of.get(id(p)) for p in params]
        if any(g is None for g in groups):
            return None
        g0 = groups[0]

        def same(key):
            return all(g.get(key) == g0.get(key) for g in groups)

        if type(opt) is torch.optim.SGD:
            if same("lr") and all(g.get("momentum", 0) == 0 and g.get("weight_decay", 0) == 0 and
                                  not g.get("nesterov", False) and not g.get("maximize", False) for g in groups):
                return ("sgd", float(g0["lr"]))
        if type(opt) is torch.optim.Adagrad:
            if same("lr") and same("eps") and all(g.get("lr_decay", 0) == 0 and g.get("weight_decay", 0) == 0 and
                                                  not g.get("maximize", False) for g in groups) and \
                    all("sum" in opt.state.get(p, {}) for p in params):
                return ("adagrad", float(g0["lr"]), float(g0["eps"]))
        if type(opt) is torch.optim.Adam:
            if same("lr") and same("eps") and same("betas") and \
                    all(g.get("weight_decay", 0) == 0 and not g.get("maximize", False) and not g.get("amsgrad", False)
                        and not g.get("capturable", False) and not g.get("fused", False) for g in groups):
                return ("adam", float(g0["lr"]), float(g0["eps"]), float(g0["betas"][0]), float(g0["betas"][1]))
        return None

    def _fused_step_state(self):
        """The fused train step applies when every piece of the step is one of our kernels: binary task with
        BCE(sum), no regulariser / auxiliary loss, tables on the fused sparse update, and every dense parameter
        is a tower weight, ``dnn_linear``, ``linear_model.weight`` or the prediction bias, all under one plain
        SGD / Adagrad.  Returns None otherwise (the step then runs through autograd + torch.optim, still on the
        GPU kernels for the lookups / interactions / tower)."""
        if os.environ.get("DCTR_FUSED_STEP", "1") == "0":
            return None
        # a user's forward hooks on the model / its prediction layer see module calls only on the stock route
        if self._forward_hooks or self._forward_pre_hooks or self.out._forward_hooks or self.out._forward_pre_hooks:
            return None
        if self._fused is not None and self._fused.get("optim") is getattr(self, "optim", None) and \
                (self._fused["slab"] is None or self._fused["slab"].intact()):
            return self._fused if self._fused["ok"] else None
        st = {"ok": False, "slab": None, "optim": getattr(self, "optim", None)}
        self._fused = st
        plan = self.model_plan()
        dnn, dnn_linear = getattr(self, "dnn", None), getattr(self, "dnn_linear", None)
        if not (getattr(self, "use_dnn", dnn is not None) and dnn is not None and dnn_linear is not None):
            return None
        if not getattr(self, "_fused_step_ok", False):    # the model's logit_parts() routes through tower_logit()
            return None
        if self.out.task != "binary" or not self.out.use_bias or self.loss_func is not F.binary_cross_entropy:
            return None
        if not plan.unit_path or plan.update[0] not in ("sgd", "adagrad", "lazy") or not plan.table_params:
            return None
        # regularisers: L2 only; on tables they belong to the lazy update (csrc/lazy.hip), on dense parameters they are
        # applied inside the slab optimizer kernel (DenseSlab.set_l2) -- never as dense autograd nodes
        table_ids = set(id(p) for p in plan.table_params)
        lam_of = {}
        for weight_list, l1, l2 in self.regularization_weight:
            if not (l1 > 0 or l2 > 0):
                continue
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                if l1 > 0:
                    return None
                if id(p) in table_ids:
                    if plan.update[0] != "lazy":
                        return None
                else:
                    lam_of[p] = lam_of.get(p, 0.0) + float(l2)
        if getattr(dnn, "dropout_rate", 0):     # the fused step is cached across train() / eval() switches
            return None
        spec = _mlp.tower_layers(dnn, dnn_linear)
        if spec is None:
            return None
        layers, w_out = spec
        known = [w for (w, b, r) in layers] + [b for (w, b, r) in layers if b is not None] + [w_out, self.out.bias]
        lw = getattr(self.linear_model, "weight", None)
        if lw is not None:
            if plan.wide_dense_weight is not lw:
                return None
            known.append(lw)
        tables = set(id(p) for p in plan.table_params)
        dense = [p for p in self.parameters() if id(p) not in tables]
        if set(id(p) for p in dense) != set(id(p) for p in known) or any(not p.requires_grad for p in dense):
            return None
        mode = self._dense_update_mode(dense)
        table_kind = plan.update[1] if plan.update[0] == "lazy" else plan.update[0]
        if mode is None or mode[0] != table_kind:
            return None
        if any(id(p) not in set(id(q) for q in dense) for p in lam_of):
            return None
        slab = _dense.DenseSlab(dense, pad_rows=[w for (w, b, r) in layers])
        if mode[0] == "adagrad":
            slab.adopt_adagrad_state(self.optim)
        elif mode[0] == "adam":
            slab.adopt_adam_state(self.optim)
        slab.set_l2(lam_of)
        slab.attach_grads()
        st.update(ok=True, slab=slab, mode=mode, one=torch.ones((), dtype=torch.float32, device=slab.flat.device))
        return st

    def _train_step_fused(self, st, xb, yb):
        slab, mode = st["slab"], st["mode"]
        plan = self.model_plan()
        # DeepFM / WDL-shaped models: the whole step as five launches enqueued straight through the C ABI, the lookup
        # inside the tower launch (_hip/step.py) -- no autograd graph, no per-step allocations
        eng = st.get("engine")
        if eng is None and getattr(self, "_gather_step", False) and _step.GatherStep.enabled():
            eng = st["engine"] = _step.GatherStep(self, slab)
        if eng is not None and _step.GatherStep.enabled() and eng.supports(xb, yb):
            loss, y_pred = eng.step(xb, yb, mode, next_xb=getattr(self, "_next_batch", None))
            slab.step(*mode)          # (applied inside the gradient kernels: clears the flag)
            return loss, loss.reshape(1), y_pred
        self._grad_sink = slab
        plan.dense_sink = slab
        # the tower's weight gradients run on a fork stream beside the embedding update (DCTR_OVERLAP_WGRAD=0: in line)
        slab.overlap = xb.is_cuda and os.environ.get("DCTR_OVERLAP_WGRAD", "1") != "0"
        # plain SGD / Adagrad: the kernels that finish the dense gradients step the parameters themselves, the
        # embedding update runs beside them on the pre-pass's stream (DCTR_INLINE_OPT=0: separate optimizer launch)
        if plan.update[0] in ("sgd", "adagrad") and os.environ.get("DCTR_INLINE_OPT", "1") != "0" and \
                plan.segments_enabled():
            slab.begin_inline_step(mode[0], mode[1], mode[2] if len(mode) > 2 else 0.0)
            topo = os.environ.get("DCTR_STEP_TOPOLOGY", "update_side")
            slab.wgrad_side = topo in ("tower_side", "tower_seg")
            slab.wgrad_on_seg = topo == "tower_seg"
            slab.gather_side = topo in ("gather_side", "flags")
            # ("flags" needs its sync block to exist before a hipGraph capture begins: a capture without an eager step
            # in front of it falls back to the event edges of "gather_side")
            slab.flag_sync = topo == "flags" and xb.is_cuda and \
                (slab._sync is not None or not torch.cuda.is_current_stream_capturing())
            if slab.flag_sync:
                slab.sync_block(xb.device)
                plan._sync_owner = slab
        reg = None
        try:
            loss, y_pred = self.fused_loss(xb, yb, slab)
            # L2 terms: part of the LOGGED loss only (their gradients are applied inside the optimizer kernels);
            # evaluated on the weights the forward used, like the reference (basemodel.py:255-257)
            reg = slab.reg_value()
            if plan.update[0] == "lazy":
                rv = plan.lazy.reg_value(xb.device)
                if rv is not None:
                    reg = rv if reg is None else reg + rv
            loss.backward(gradient=st["one"])       # a resident 1.0: no fill launch per step
        except BaseException:
            if slab.flag_sync and not torch.cuda.is_current_stream_capturing():
                try:                 # a step interrupted between a signal and its wait: start the pairs over
                    slab.check_sync(reset=True)
                except RuntimeError:
                    pass
            raise
        finally:
            self._grad_sink = None
            plan.dense_sink = None
            slab.overlap = False
            after = getattr(slab, "after_update", None)
            if after is not None:         # (no embedding update was launched behind the tower: fork now)
                slab.after_update = None
                after()
            slab.end_inline_step()
            if not ((slab.wgrad_side or slab.gather_side) and getattr(self, "_defer_dense_join", False)):
                # (inside a multi-step hipGraph the captured steps but the last leave the forked weight-gradient /
                # optimizer kernels unjoined: the next step's tower launch is the first reader of what they write)
                slab.join()
        slab.step(*mode)
        total = loss.detach().reshape(1)
        if reg is not None:
            total = total + reg
        return loss.detach(), total, y_pred

    def _train_step(self, xb, yb):
        """forward -> loss(sum) + reg + aux -> backward (fused sparse update inside) -> dense optimizer step
        (reference basemodel.py:242-262).  Returns device tensors; nothing is synchronised."""
        # (not tied to self.training: like the reference, fit() keeps training in eval mode after the first validation
        # pass, basemodel.py:215,331; the fused step has no mode-dependent layer -- dropout / BatchNorm rule it out)
        self._sync_optimizer_hyper()
        if self._aux_is_default():
            st = self._fused_step_state()
            if st is not None:
                return self._train_step_fused(st, xb, yb)
        l2map = None      # {id(param): (param, lambda)}: L2 terms applied by the optimizer kernel instead of autograd
        lazy_reg = None
        parts = self.logit_parts(xb) if self._bce_head_ok(xb) else None
        if parts is not None and not (0 < len(parts) <= 4 and yb.numel() == xb.shape[0] and all(
                torch.is_tensor(q) and q.numel() == xb.shape[0] for q in parts)):
            logit = parts[0]                       # (something the fused head does not take: BaseModel.forward's sum)
            for q in parts[1:]:
                logit = logit + q
            y_pred, parts = self.out(logit).squeeze(), None
            self.optim.zero_grad()
            loss = self.loss_func(y_pred, yb.squeeze(), reduction='sum')
            total_loss = loss + self.get_regularization_loss() + self.aux_loss
        elif parts is not None:
            # binary task, BCE(sum): logit adds, PredictionLayer, loss and their backward as ONE launch
            # (csrc/head.hip k_bce_head) instead of ~15 elementwise / reduce launches of pure latency
            loss, y_pred = _mlp.bce_head(parts, self.out.bias if self.out.use_bias else None, yb, unit=True)
            self.optim.zero_grad()
            total_loss = loss.reshape(1)
            if self._has_reg_terms():
                l2map = self._fusable_l2()
                if l2map is None:
                    total_loss = total_loss + self.get_regularization_loss()
                elif self._plan is not None and self._plan.update[0] == "lazy":
                    # the lazily regularised tables' term of the logged loss, on the weights the forward used
                    lazy_reg = self._plan.lazy.reg_value(self.device)
            if not self._aux_is_default():
                total_loss = total_loss + self.aux_loss
        else:
            y_pred = self(xb).squeeze()
            self.optim.zero_grad()
            if isinstance(self.loss_func, list):
                assert len(self.loss_func) == self.num_tasks, \
                    "the length of `loss_func` should be equal with `self.num_tasks`"
                loss = sum([self.loss_func[i](y_pred[:, i], yb[:, i], reduction='sum') for i in range(self.num_tasks)])
            else:
                loss = self.loss_func(y_pred, yb.squeeze(), reduction='sum')
            total_loss = loss + self.get_regularization_loss() + self.aux_loss
        total_loss.backward()
        self._step_stacked_groups()
        done, reg = self._step_dense_multi(l2map)
        if not done:
            self.optim.step()
        total = total_loss.detach()
        if reg is not None:       # the fused L2 terms (+ the lazily regularised tables') enter the LOGGED loss only
            total = total + reg
        if lazy_reg is not None:
            total = total + lazy_reg
        return loss.detach(), total, y_pred.detach()

    def _fusable_l2(self):
        """``{id(p): (p, lambda)}`` when every regularisation term of the model is a plain L2 term on a dense parameter
        that ``_step_dense_multi`` is certain to step (one term per parameter; fp32, contiguous, on the GPU; the whole
        optimizer a plain SGD / Adagrad with one lr): the kernel then adds ``2 lambda p`` to the gradient itself -- for
        DCN's default ``l2_reg_cross`` the autograd route is ~40 launches of pow / mul / sum / add per step.  ``None``
        when anything else is regularised (L1, a table on the dense-gradient route, a stacked weight group, ...)."""
        if os.environ.get("DCTR_MULTI_STEP", "1") == "0" or os.environ.get("DCTR_FUSED_L2", "1") == "0":
            return None
        every = [p for grp in self.optim.param_groups for p in grp["params"]]
        mode = self._dense_update_mode(every)
        if mode is None or mode[0] not in ("sgd", "adagrad"):
            return None
        lazy_ids = set()
        if self._plan is not None and self._plan.update[0] == "lazy":
            lazy_ids = set(id(p) for p in self._plan.table_params)
        stacked = set()
        for mod in self.modules():
            fn = getattr(mod, "stacked_weights", None)
            sw = fn() if fn is not None else None
            if sw is not None:
                stacked.update(id(w) for w in sw[0])
        known = set(id(p) for p in every)
        out = {}
        for weight_list, l1, l2 in self.regularization_weight:
            if not (l1 > 0 or l2 > 0):
                continue
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                if id(p) in lazy_ids:
                    continue
                if l1 > 0 or id(p) in out or id(p) in stacked or id(p) not in known or not p.is_cuda or \
                        p.dtype != torch.float32 or not p.is_contiguous() or not p.requires_grad:
                    return None
                out[id(p)] = (p, float(l2))
        return out

    def _bce_head_ok(self, xb):
        """True when the autograd-route step may take the fused prediction head: the reference's binary task --
        ``self.out`` a sigmoid PredictionLayer, ``F.binary_cross_entropy`` (basemodel.py:254, 464-477) -- on a model
        whose ``forward`` is the stock sum of ``logit_parts``."""
        if not xb.is_cuda or os.environ.get("DCTR_FUSED_HEAD", "1") == "0":
            return False
        from ..layers.core import PredictionLayer
        if self.loss_func is not F.binary_cross_entropy or type(self.out) is not PredictionLayer or \
                self.out.task != "binary" or getattr(self, "num_tasks", 1) != 1:
            return False
        # (a user's hooks on the model or on its prediction layer must keep firing: they see the module calls only on the
        # stock route)
        if self._forward_hooks or self._forward_pre_hooks or self.out._forward_hooks or self.out._forward_pre_hooks:
            return False
        cls = type(self)
        return cls.forward is BaseModel.forward and cls.logit_parts is not BaseModel.logit_parts

    def _has_reg_terms(self):
        """False when ``get_regularization_loss()`` is identically zero (no term with a positive strength, no lazily
        regularised table): the logged total loss is then the loss itself and three launches are saved."""
        if self._plan is not None and self._plan.update[0] == "lazy":
            return True
        return any((l1 > 0 or l2 > 0) and len(weight_list) > 0 for weight_list, l1, l2 in self.regularization_weight)

    def _step_dense_multi(self, l2map=None):
        """One ``dctr_dense_opt_multi`` launch for every dense parameter autograd left a gradient on, when the compiled
        optimizer is a plain SGD / Adagrad over them (``_dense_update_mode``): ``torch.optim``'s foreach walk is five
        launches, 75-80 us per xDeepFM / FiBiNET / DCN step.  Returns True when it stepped EVERY such parameter
        (``optim.step()`` then has nothing to do; ``p.grad`` stays visible like the reference's, EXCEPT that a fused L2
        term -- ``l2map`` -- is added inside the kernel and not written back: ``p.grad`` then lacks the reference's
        ``2 lambda p``); parameters it cannot take (non-contiguous, another dtype / device) are left to ``optim.step()``
        with the stepped ones hidden.  Skipping ``optim.step()`` keeps its bookkeeping honest by hand: Adagrad's per-parameter
        ``state['step']`` and the ``_step_count`` that learning-rate schedulers check are advanced here."""
        if os.environ.get("DCTR_MULTI_STEP", "1") == "0":
            return False, None
        todo, rest = [], []
        for grp in self.optim.param_groups:
            for p in grp["params"]:
                g = p.grad
                if g is None and l2map and id(p) in l2map:
                    g = p.grad = torch.zeros_like(p)      # (autograd would have left 2 lambda p here)
                if g is None:
                    continue
                if p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and \
                        g.is_contiguous() and not g.is_sparse and g.device == p.device:
                    todo.append(p)
                else:
                    rest.append(p)
        if l2map and any(id(p) in l2map for p in rest):
            raise RuntimeError("an L2-regularised parameter changed its layout during the step")
        if not todo:
            return False, None
        mode = self._dense_update_mode(todo)
        if mode is None or mode[0] not in ("sgd", "adagrad"):
            if l2map:
                raise RuntimeError("the optimizer changed under a train step with fused L2 terms")
            return False, None
        import ctypes
        from .._hip import lib as L
        if mode[0] == "adagrad":
            sums = [self.optim.state[p]["sum"] for p in todo]
            if any(s.dtype != torch.float32 or not s.is_contiguous() or s.device != p.device or s.shape != p.shape
                   for s, p in zip(sums, todo)):
                if l2map:
                    raise RuntimeError("Adagrad state of an L2-regularised parameter is not a plain fp32 tensor")
                return False, None
        items = (L.DenseItem * len(todo))()
        for i, p in enumerate(todo):
            items[i].p, items[i].g, items[i].n = p.data_ptr(), p.grad.data_ptr(), p.numel()
            items[i].state = sums[i].data_ptr() if mode[0] == "adagrad" else None
            items[i].l2 = l2map[id(p)][1] if (l2map and id(p) in l2map) else 0.0
        stream = L.stream_handle(todo[0].device)
        reg = None
        if l2map:
            # value of the fused terms on the weights the forward used (before the step), for the logged loss
            reg = torch.empty((1,), dtype=torch.float32, device=todo[0].device)
            L.check(L.lib().dctr_l2_value_multi(items, len(todo), ctypes.c_void_p(reg.data_ptr()), stream),
                    "dctr_l2_value_multi")
        L.check(L.lib().dctr_dense_opt_multi(items, len(todo), L.UPD_ADAGRAD if mode[0] == "adagrad" else L.UPD_SGD,
                                             float(mode[1]), float(mode[2]) if len(mode) > 2 else 0.0, stream),
                "dctr_dense_opt_multi")
        # the bookkeeping torch.optim would have done for the parameters stepped HERE -- also when `rest` goes on to
        # optim.step() (round-3 advisor: the early return left Adagrad's state['step'] of `todo` behind)
        if mode[0] == "adagrad":
            for p in todo:
                st = self.optim.state[p].get("step")
                if torch.is_tensor(st):
                    st += 1
                elif st is not None:
                    self.optim.state[p]["step"] = st + 1
        if rest:
            for p in todo:
                p.grad = None
            return False, reg
        if hasattr(self.optim, "_step_count"):
            self.optim._step_count += 1       # (what older torch.optim.lr_scheduler checks to warn about a skipped step())
        self.optim._opt_called = True         # (... and what newer ones check)
        return True, reg

    def _step_stacked_groups(self):
        """Layers that keep many small parameters as slices of one slab (FiBiNET's 2 x 325 bilinear ``nn.Linear``
        weights, whose gradients the kernels write as one ``[n_w, D, D]`` tensor as well) are stepped with ONE
        ``dctr_dense_opt`` launch per group here -- ``torch.optim``'s foreach kernels over 655 tensors were 30 launches,
        161 us per FiBiNET step -- and then hidden from ``optim.step()`` (gradient None: torch skips them).  Only for a
        plain SGD / Adagrad over the group (the same condition as the fused dense step); the optimizer's ``sum`` state
        of the group is re-seated once as slices of a state slab, so ``optimizer.state_dict()`` keeps working."""
        if os.environ.get("DCTR_STACKED_STEP", "1") == "0":
            return
        groups = self.__dict__.get("_stacked_cache")
        if groups is None or groups[0] is not self.optim:
            found = []
            for mod in self.modules():
                fn = getattr(mod, "stacked_weights", None)
                if fn is not None:
                    found.append(mod)
            groups = self._stacked_cache = (self.optim, found, {})
        _, mods, states = groups
        if not mods:
            return
        import ctypes
        from .._hip import lib as L
        for mod in mods:
            sw = mod.stacked_weights()
            if sw is None:
                continue
            ws, slab = sw
            g0 = ws[0].grad
            if g0 is None or not slab.is_cuda:
                continue
            mode = self._dense_update_mode(ws)
            if mode is None or mode[0] not in ("sgd", "adagrad"):
                continue
            n, step = slab.numel(), slab[0].numel() * 4
            gl = ws[-1].grad
            if gl is None or g0.dtype != torch.float32 or not g0.is_contiguous() or \
                    gl.data_ptr() != g0.data_ptr() + (len(ws) - 1) * step:
                continue          # the gradients are not the kernels' single [n_w, D, D] tensor: leave it to torch
            st = None
            if mode[0] == "adagrad":
                st = states.get(id(mod))
                if st is None or st.shape != slab.shape or st.device != slab.device or \
                        self.optim.state[ws[0]]["sum"].data_ptr() != st.data_ptr():
                    st = torch.stack([self.optim.state[w]["sum"] for w in ws]).contiguous()
                    for i, w in enumerate(ws):
                        self.optim.state[w]["sum"] = st[i]
                    states[id(mod)] = st
            L.check(L.lib().dctr_dense_opt(ctypes.c_void_p(slab.data_ptr()), ctypes.c_void_p(g0.data_ptr()),
                                           ctypes.c_void_p(st.data_ptr()) if st is not None else None, n,
                                           L.UPD_ADAGRAD if mode[0] == "adagrad" else L.UPD_SGD, float(mode[1]),
                                           float(mode[2]) if len(mode) > 2 else 0.0, L.stream_handle(slab.device)),
                    "dctr_dense_opt(stacked group)")
            for w in ws:
                w.grad = None

    def _graph_safe_step(self):
        """True when a hipGraph replay of ``_train_step`` does what an eager call does: no host-side value that changes
        from step to step may be baked into a launch.  The fused step qualifies (its Adam step count lives on the
        device).  The autograd step qualifies when the tables are updated by our kernels (not the "dense" mode, whose
        gradient-slab bookkeeping is host-side) and the dense parameters are under plain SGD or Adagrad without
        lr_decay -- torch's non-capturable Adam computes its bias corrections on the host, a replay would freeze them."""
        if self._fused is not None and self._fused.get("ok"):
            return True
        if not self._aux_is_default():
            return False
        plan = self._plan
        if plan is None or plan.update[0] == "dense" or not plan.table_params:
            return False
        opt = getattr(self, "optim", None)
        if type(opt) is torch.optim.SGD:
            return True
        if type(opt) is torch.optim.Adagrad:
            return all(g.get("lr_decay", 0) == 0 for g in opt.param_groups)
        return False

    def _fit_group_size(self, n_full, on_gpu):
        """Train steps per hipGraph inside ``fit()``: a graph launch leaves the GPU idle for ~12 us, a step is 85-100 us,
        so whole GROUPS of consecutive full-size batches replay as one graph (DCTR_FIT_STEPS_PER_GRAPH, default 16) when
        an epoch holds enough of them; 1 = one step per graph (small datasets)."""
        S = int(os.environ.get("DCTR_FIT_STEPS_PER_GRAPH", "16"))
        if not on_gpu or os.environ.get("DCTR_FIT_GRAPH", "1") == "0" or S <= 1 or n_full < S + 2:
            return 1
        return S

    def _fit_graph_ready(self, xb_like, S):
        g = self._fit_graph
        if g is None or g.get("graph") is None or g["fused"] is not self._fused:
            return None
        gr = g["graph"]
        return gr if (gr.S == S and gr.valid_for(xb_like)) else None

    def _fit_step(self, xb, yb, batch_size, S=1):
        """One training step of ``fit``: full-size batches replay a hipGraph of the train step (one launch per step
        instead of 10-100 kernel launches through Python) whenever the step is replay-safe (``_graph_safe_step``);
        everything else -- the ragged last batch, steps that bake host-side values into their launches, CPU-side
        debugging with DCTR_FIT_GRAPH=0 -- runs ``_train_step`` directly.  Same arithmetic either way.  ``S`` > 1: the
        captured graph holds S steps and is replayed by ``fit`` itself on whole groups of rows
        (``GraphedTrainStep.step_rows``); a single step beside the groups (warm-up, tail) runs eagerly."""
        self._sync_optimizer_hyper(full=False)      # (drops a captured step whose launches carry the old values)
        g = self._fit_graph
        if xb.shape[0] != batch_size or not xb.is_cuda or os.environ.get("DCTR_FIT_GRAPH", "1") == "0":
            return self._train_step(xb, yb)
        if g is not None and g["graph"] is not None and g["graph"].valid_for(xb) and g["fused"] is self._fused and \
                g["graph"].S == S:
            return g["graph"](xb, yb) if S == 1 else self._train_step(xb, yb)
        if g is None or g.get("shape") != tuple(xb.shape) or g["fused"] is not self._fused or g.get("S", 1) != S:
            g = self._fit_graph = {"shape": tuple(xb.shape), "warm": 0, "graph": None, "fused": self._fused, "S": S}
        out = self._train_step(xb, yb)              # eager warm-up steps (also builds the fused-step state)
        g["warm"] += 1
        g["fused"] = self._fused
        if g["warm"] >= 2 and self._graph_safe_step():
            from .._hip.graph import GraphedTrainStep
            try:
                g["graph"] = GraphedTrainStep(self, xb, yb, steps_per_graph=S, inputs_ready=S > 1).capture(xb, yb)
            except Exception as e:                 # capture is an optimisation; keep training eagerly
                import warnings
                warnings.warn("fit(): hipGraph capture of the train step failed (%s: %s); training eagerly"
                              % (type(e).__name__, e))
                g["graph"] = None
                g["warm"] = -10 ** 9
                torch.cuda.synchronize()
        return out

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, initial_epoch=0, validation_split=0.,
            validation_data=None, shuffle=True, callbacks=None):
        """Same contract as the reference (basemodel.py:137-309); returns ``self.history``."""
        if isinstance(x, dict):
            x = [x[feature] for feature in self.feature_index]
        do_validation = False
        val_x, val_y = [], []
        if validation_data:
            do_validation = True
            if len(validation_data) == 2:
                val_x, val_y = validation_data
            elif len(validation_data) == 3:
                val_x, val_y, _ = validation_data
            else:
                raise ValueError('When passing a `validation_data` argument, it must contain either 2 items '
                                 '(x_val, y_val), or 3 items (x_val, y_val, val_sample_weights). '
                                 'However we received `validation_data=%s`' % (validation_data,))
            if isinstance(val_x, dict):
                val_x = [val_x[feature] for feature in self.feature_index]
        elif validation_split and 0. < validation_split < 1.:
            do_validation = True
            if torch.is_tensor(x) and x.dim() == 2:
                # a resident [N, sum(widths)] matrix (see _as_matrix): split its ROWS (x[0] is a row here, not a feature)
                split_at = int(x.shape[0] * (1. - validation_split))
                x, val_x = x[:split_at], x[split_at:]
            else:
                n0 = x[0].shape[0] if hasattr(x[0], 'shape') else len(x[0])
                split_at = int(n0 * (1. - validation_split))
                x, val_x = slice_arrays(x, 0, split_at), slice_arrays(x, split_at)
            y, val_y = slice_arrays(y, 0, split_at), slice_arrays(y, split_at)

        self._sync_optimizer_hyper()
        X_all = self._as_matrix(x)                                   # resident in HBM for the whole fit
        y_all = y.to(self.device).float() if torch.is_tensor(y) else torch.from_numpy(np.asarray(y)).to(self.device).float()
        if batch_size is None:
            batch_size = 256
        self.train()
        from .. import distributed_fit as _dfit
        if _dfit.context() is not None:
            # one process per GPU (torchrun): the minibatch is sharded over the ranks, batch_size is per GPU like the
            # reference's `batch_size *= len(gpus)` under nn.DataParallel (basemodel.py:206-209)
            return _dfit.fit(self, X_all, y_all, batch_size, epochs, verbose, initial_epoch, do_validation, val_x, val_y,
                             shuffle, callbacks)
        if self.gpus:
            print('parallel running on these gpus:', self.gpus,
                  '-- nn.DataParallel is not used; launch one process per GPU with torchrun: fit() then shards the '
                  'minibatch over the ranks (deepctr_torch/distributed_fit.py)')
        else:
            print(self.device)
        sample_num = X_all.shape[0]
        steps_per_epoch = (sample_num - 1) // batch_size + 1

        cbs = _cb.CallbackList((callbacks or []) + [self.history])
        cbs.set_model(self)
        cbs.on_train_begin()
        cbs.set_model(self)
        self.stop_training = False

        print("Train on {0} samples, validate on {1} samples, {2} steps per epoch".format(
            sample_num, len(val_y), steps_per_epoch))
        plan = self.model_plan()
        n_full = sample_num // batch_size
        S = self._fit_group_size(n_full, X_all.is_cuda and X_all.device.type == "cuda")

        def draw_order():
            # The reference iterates a DataLoader (:213,240): each epoch its iterator first draws a base seed from the
            # default CPU generator, then (shuffle=True) RandomSampler draws the seed of its permutation.  The same two
            # draws here: after torch.manual_seed(s) both implementations visit the rows in the same order.
            torch.empty((), dtype=torch.int64).random_()
            if not shuffle:
                return None
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
            return torch.randperm(sample_num, generator=gen).to(self.device)

        # The permutations of the epochs AHEAD are drawn on worker threads while the GPU runs (torch.randperm on the host: 11 ms
        # per 262 144 rows on the benchmark box, against 6 ms of GPU work for such an epoch -- fit() of short epochs was
        # host-bound at 0.46 ms per step, profiles/r05_fit_profile.txt) -- only when nothing else can draw from the default
        # generator in between (no validation pass, no user callbacks): the two draws per epoch the reference makes are then
        # made up front, in the reference's order, so the generator ends where the reference's ends; every permutation comes
        # from its own seeded generator, as in RandomSampler.
        lookahead = shuffle and not do_validation and not callbacks and S > 1
        perm_pool, perm_futs = None, {}
        if lookahead and epochs - initial_epoch > 1:
            from concurrent.futures import ThreadPoolExecutor
            seeds = []
            for _ in range(initial_epoch, epochs):
                torch.empty((), dtype=torch.int64).random_()
                seeds.append(int(torch.empty((), dtype=torch.int64).random_().item()))
            perm_pool = ThreadPoolExecutor(max_workers=min(4, len(seeds)))

            def _perm(seed):
                gen = torch.Generator()
                gen.manual_seed(seed)
                return torch.randperm(sample_num, generator=gen)

            def _submit(e):       # at most four permutations in flight / parked (8 bytes x rows each)
                if e < epochs and e not in perm_futs:
                    perm_futs[e] = perm_pool.submit(_perm, seeds[e - initial_epoch])
            for e in range(initial_epoch, min(epochs, initial_epoch + 4)):
                _submit(e)
        for epoch in range(initial_epoch, epochs):
            cbs.on_epoch_begin(epoch)
            epoch_logs = {}
            start_time = time.time()
            if perm_pool is not None:
                order = perm_futs.pop(epoch).result().to(self.device)
                _submit(epoch + 4)
            else:
                order = draw_order()
            total_acc = torch.zeros((), device=self.device, dtype=torch.float64)
            preds = [] if (verbose > 0 and self.metrics) else None
            bar = tqdm(total=steps_per_epoch, disable=verbose != 1) if tqdm is not None else None
            try:
                step, synced = 0, False
                while step < steps_per_epoch:
                    gr = self._fit_graph_ready(X_all[:batch_size], S) if (S > 1 and step + S <= n_full) else None
                    if gr is not None:
                        # S consecutive full-size batches: one index_select pair + one graph launch for the group
                        self._sync_optimizer_hyper(full=False)
                        gr = self._fit_graph_ready(X_all[:batch_size], S)
                    if gr is not None:
                        if not synced:
                            gr.sync_inputs()            # this epoch's permutation is complete before the first gather
                            synced = True
                        lo = step * batch_size
                        outs = gr.step_rows(X_all, y_all, lo, order)
                        tl = [o[1] for o in outs]
                        if all(t.numel() == 1 for t in tl):
                            total_acc += torch.stack([t.reshape(()) for t in tl]).double().sum()
                        else:
                            for t in tl:
                                total_acc += t.double().sum()
                        if preds is not None:
                            for j, o in enumerate(outs):
                                a, b = lo + j * batch_size, lo + (j + 1) * batch_size
                                yb = y_all.index_select(0, order[a:b]) if order is not None else y_all[a:b]
                                preds.append((yb, o[2].clone()))
                        if bar is not None:
                            bar.update(S)
                        step += S
                        continue
                    lo, hi = step * batch_size, min((step + 1) * batch_size, sample_num)
                    if order is not None:
                        idx = order[lo:hi]
                        xb, yb = X_all.index_select(0, idx), y_all.index_select(0, idx)
                    else:
                        xb, yb = X_all[lo:hi], y_all[lo:hi]
                    loss, total_loss, y_pred = self._fit_step(xb, yb, batch_size, S)
                    # one launch: fp64 += fp32 promotes inside the add (cast + sum + add were three)
                    total_acc += total_loss.reshape(()) if total_loss.numel() == 1 else total_loss.double().sum()
                    if preds is not None:
                        preds.append((yb, y_pred.clone()))
                    if bar is not None:
                        bar.update(1)
                    step += 1
            finally:
                if bar is not None:
                    bar.close()
            plan.check_ids()
            epoch_logs["loss"] = float(total_acc.item()) / sample_num
            if preds is not None:
                # reference: metric of every batch, averaged over steps (basemodel.py:264-269,280)
                per_batch = {name: [] for name in self.metrics}
                for yb, y_pred in preds:
                    yt, yp = yb.cpu().numpy(), y_pred.cpu().numpy().astype("float64")
                    for name, fun in self.metrics.items():
                        per_batch[name].append(fun(yt, yp))
                for name, vals in per_batch.items():
                    epoch_logs[name] = np.sum(vals) / steps_per_epoch
            if do_validation:
                for name, result in self.evaluate(val_x, val_y, batch_size).items():
                    epoch_logs["val_" + name] = result
            if verbose > 0:
                epoch_time = int(time.time() - start_time)
                print('Epoch {0}/{1}'.format(epoch + 1, epochs))
                eval_str = "{0}s - loss: {1: .4f}".format(epoch_time, epoch_logs["loss"])
                for name in self.metrics:
                    eval_str += " - " + name + ": {0: .4f}".format(epoch_logs[name])
                if do_validation:
                    for name in self.metrics:
                        eval_str += " - " + "val_" + name + ": {0: .4f}".format(epoch_logs["val_" + name])
                print(eval_str)
            self._flush_lazy()       # callbacks (and whoever reads .weight after fit) see the reference's tables
            cbs.on_epoch_end(epoch, epoch_logs)
            if self.stop_training:
                break
        if perm_pool is not None:
            perm_pool.shutdown(wait=False)
        cbs.on_train_end()
        return self.history

    def evaluate(self, x, y, batch_size=256):
        pred_ans = self.predict(x, batch_size)
        return {name: fun(y, pred_ans) for name, fun in self.metrics.items()}

    def predict(self, x, batch_size=256):
        """float64 ``[N, 1]`` predictions, input order preserved (reference basemodel.py:325-352)."""
        self.eval()  # like the reference (:331), predict leaves the model in eval mode
        X_all = self._as_matrix(x)
        if self.__dict__.get("optim") is None and X_all.is_cuda and self.model_plan() is not None:
            # never compiled for training (weights loaded, then predict / evaluate): the forward-only table layout -- deep
            # row and wide weight of an id in one 128-byte line (_hip/layout.py)
            from .._hip.layout import apply_infer_layout
            apply_infer_layout(self._plan)
        torch.empty((), dtype=torch.int64).random_()     # the base seed the reference's DataLoader iterator draws (:340)
        # In eval mode every sample's prediction is independent of its batch (no batch statistics, no dropout), so the
        # caller's batch_size -- 256 by default in the reference, i.e. ~4 launches per 256 rows -- only sets a lower
        # bound on the rows per launch here: identical values, far fewer launches.
        step = max(int(batch_size), int(os.environ.get("DCTR_PREDICT_ROWS", "8192")))
        chunks = []
        with torch.no_grad():
            for lo in range(0, X_all.shape[0], step):
                chunks.append(self(X_all[lo:lo + step]))
        if self._plan is not None:
            self._plan.check_ids()
        return torch.cat(chunks).cpu().numpy().astype("float64")
