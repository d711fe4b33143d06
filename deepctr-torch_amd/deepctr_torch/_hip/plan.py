"""Feature columns -> ``dctr_plan_t``: the compiled schema the gather / scatter kernels run on.

One :class:`EmbeddingPlan` plays the role of everything ``build_input_features``,
``create_embedding_matrix`` and ``Linear.__init__`` set up in the reference (inputs.py:99-180,
basemodel.py:34-61): which X column feeds which table, how a VarLen feature is pooled, and where each
field lands in the DNN-input row (``combined_dnn_input`` order, inputs.py:126-138).

HBM layout (sized for 288 GB): every table is its own ``[V, D]`` fp32 tensor (the ``nn.Embedding``
weight, so ``state_dict`` keys stay the reference's); next to a table may sit two same-shaped slabs,
``gacc`` (gradient accumulator, ZERO AT REST: the scatter kernel adds into it and the optimizer pass
re-zeroes exactly the rows it consumed, so no O(V) memset ever runs) and ``state`` (Adagrad sum).
"""
import ctypes
import os
import weakref

import torch

from . import lib as L
from . import streams as _streams
from ..inputs import DenseFeat, SparseFeat, VarLenSparseFeat, split_columns


# Slabs are keyed by the table Parameter itself (not by plan): every plan over a table shares them, they
# die with the parameter, and they never travel inside ``torch.save(model)``.
class _ParamMap(object):
    """Weak map Parameter -> tensor keyed by identity (tensor ``==`` is elementwise, which rules out
    ``weakref.WeakKeyDictionary``)."""

    def __init__(self):
        self._d = {}

    def get(self, p, default=None):
        hit = self._d.get(id(p))
        return hit[1] if hit is not None and hit[0]() is p else default

    def __contains__(self, p):
        return self.get(p) is not None

    def __getitem__(self, p):
        v = self.get(p)
        if v is None:
            raise KeyError("no slab for this parameter")
        return v

    def __setitem__(self, p, value):
        key = id(p)
        self._d[key] = (weakref.ref(p, lambda _r, k=key, d=self._d: d.pop(k, None)), value)

    def __delitem__(self, p):
        self._d.pop(id(p), None)


_GACC = _ParamMap()
_STATE = _ParamMap()


class _FieldSpec(object):
    __slots__ = ("name", "param", "col", "len", "pool", "len_col", "out_off", "dim", "vocab")

    def __init__(self, name, param, col, length, pool, len_col, out_off):
        self.name, self.param, self.col, self.len = name, param, col, length
        self.pool, self.len_col, self.out_off = pool, len_col, out_off
        self.vocab, self.dim = int(param.shape[0]), int(param.shape[1])


def _fields_for(columns, tables, feature_index, unpooled):
    """Fixed-length fields first, then VarLen ones (the order of
    ``sparse_embedding_list + varlen_sparse_embedding_list``, basemodel.py:380)."""
    sparse_cols, varlen_cols, _ = split_columns(columns)
    fixed, pooled = [], []
    off = 0
    for fc in sparse_cols:
        w = tables[fc.embedding_name].weight
        fixed.append(_FieldSpec(fc.name, w, feature_index[fc.name][0], 1, 0, -1, off))
        off += int(w.shape[1])
    for fc in varlen_cols:
        w = tables[fc.embedding_name].weight
        lo, hi = feature_index[fc.name]
        if unpooled:  # [B, maxlen, D]: every position is its own fixed field
            for t in range(hi - lo):
                fixed.append(_FieldSpec("%s[%d]" % (fc.name, t), w, lo + t, 1, 0, -1, off))
                off += int(w.shape[1])
        else:
            if fc.combiner not in ("sum", "mean", "max"):
                raise ValueError('parameter mode should in [sum, mean, max]')
            len_col = -1 if fc.length_name is None else feature_index[fc.length_name][0]
            pooled.append(_FieldSpec(fc.name, w, lo, hi - lo, L.POOL_CODE[fc.combiner], len_col, off))
            off += int(w.shape[1])
    return fixed, pooled, off


class LazyState(object):
    """Host side of csrc/lazy.hip: the reference's regularised / Adam table update replayed lazily, step by step.

    ``l2`` maps a table parameter to its lambda (0 when unregularised), ``s1`` / ``s2`` to its optimizer state
    tensors (Adagrad ``sum`` | RMSprop ``square_avg`` | Adam ``exp_avg``, Adam ``exp_avg_sq``).  Owns the per-unit stamps, the device step
    counter and the gradient slabs' use; see include/dctr.h for the protocol."""

    # tables up to this many elements in total get their logged regularisation term recomputed EXACTLY at every
    # step (flush + one reduction, cheap when small); bigger models log the value of the last flush
    EXACT_REG_ELEMS = 1 << 22

    def __init__(self, plan, kind, lr, eps, beta1, beta2, l2, s1, s2, optimizer=None):
        self.plan, self.kind = plan, kind
        self.hyper = (float(lr), float(eps), float(beta1), float(beta2))
        self.l2 = dict((id(p), float(v)) for p, v in l2.items())
        self.s1 = dict((id(p), v) for p, v in (s1 or {}).items())
        self.s2 = dict((id(p), v) for p, v in (s2 or {}).items())
        self.optimizer = optimizer
        self.stamps = None
        self.step = None
        self.dirty = False
        self._units_dev = None
        self._key = None
        self._reg = None          # value for the current weights (dropped by apply())
        self._last_reg = None     # last value computed (what big models log between two flushes)
        self.opt = L.LazyOpt()
        self.opt.kind = {"sgd": L.LAZY_SGD, "adagrad": L.LAZY_ADAGRAD, "adam": L.LAZY_ADAM,
                         "rmsprop": L.LAZY_RMSPROP}[kind]
        self.opt.lr, self.opt.eps, self.opt.beta1, self.opt.beta2 = self.hyper
        self.opt.any_l2 = int(any(v > 0 for v in self.l2.values()))
        # rows move between two touches (an L2 term, or moments that keep decaying): there are steps to replay
        self.replays = bool(self.opt.any_l2) or kind in ("adam", "rmsprop")
        # every train step also brings one K-th of every table to the current step (dctr_lazy_sweep): no row sleeps longer
        # than K steps (0: off -- rows pay their whole history when they are next drawn or at the next flush)
        self.sweep_k = int(os.environ.get("DCTR_LAZY_SWEEP_K", "32"))
        self._sweep_side = None   # the side stream a sweep was forked on and not yet joined (see _fork_sweep)
        self.vec = 4 if plan.vec == 4 else 1
        self.max_dim = max(plan.max_dim, 1)
        self.n_elems = sum(p.numel() for p in plan.table_params)
        self._adam_tab = None     # (ss, bc) device tables of Adam's step-dependent scalars (include/dctr.h dctr_lazy_opt_t)

    def signature(self):
        return (self.kind, self.hyper, tuple(sorted(self.l2.items())))

    ADAM_TABLE_MAX = 1 << 22

    @staticmethod
    def adam_tables(lr, beta1, beta2, limit=1 << 22):
        """(step_size[T - 1], sqrt(bias_correction2)[T - 1]) for T = 1 .. n as float32 arrays, computed like torch.optim.Adam
        computes them per step on the host (adam.py: ``1 - beta ** step`` in double, ``lr / bias_correction1``,
        ``sqrt(bias_correction2)``), each up to the first step whose bias correction is exactly 1 in double (the value
        from there on); None when a table would exceed ``limit`` entries (a beta within 1e-5 of 1)."""
        import math
        import numpy as np
        out = []
        for beta in (float(beta1), float(beta2)):
            if not 0.0 <= beta < 1.0:
                return None
            n = 1 if beta == 0.0 else int(math.ceil(math.log(2.0 ** -54) / math.log(beta))) + 2
            if n > limit:
                return None
            corr = 1.0 - np.power(np.float64(beta), np.arange(1, n + 1, dtype=np.float64))
            if corr[-1] != 1.0:
                return None
            out.append(corr)
        ss = (np.float64(lr) / out[0]).astype(np.float32)
        bc = np.sqrt(out[1]).astype(np.float32)
        rbc = (1.0 / np.sqrt(out[1])).astype(np.float32)       # (the replay loop's divisor as a factor: dctr_lazy_opt_t.adam_rbc)
        return ss, bc, rbc

    def _ensure_adam_tables(self, device):
        if self.kind != "adam" or os.environ.get("DCTR_LAZY_ADAM_TABLES", "1") == "0":
            return
        if self._adam_tab is None or self._adam_tab[0].device != torch.device(device):
            # (from the hyper-parameters as the optimizer holds them, in double -- the in-kernel fallback starts from their
            # float32 roundings: beta2 = 0.999 is 1.3e-8 off there, 4e-6 of the step size after a thousand steps)
            lr, _, b1, b2 = self.hyper
            tabs = self.adam_tables(lr, b1, b2, self.ADAM_TABLE_MAX)
            if tabs is None:
                self._adam_tab = ()
            else:
                self._adam_tab = tuple(torch.from_numpy(t).to(device) for t in tabs)
        if self._adam_tab:
            ss, bc, rbc = self._adam_tab
            self.opt.adam_ss, self.opt.n_ss = ss.data_ptr(), ss.numel()
            self.opt.adam_bc, self.opt.n_bc = bc.data_ptr(), bc.numel()
            self.opt.adam_rbc = rbc.data_ptr()

    def _ensure(self, device):
        plan = self.plan
        if self.step is None or self.step.device != torch.device(device):
            t0 = 0
            if self.optimizer is not None and self.kind in ("adam", "rmsprop"):   # resume: Adam's bias correction needs t;
                # (RMSprop's step only counts, but flush() writes the device counter back into optimizer.state)
                for p in plan.table_params:
                    st = self.optimizer.state.get(p, {})
                    if "step" in st:
                        t0 = max(t0, int(float(st["step"])))
            self.step = torch.full((1,), t0, dtype=torch.int32, device=device)
            self.stamps = [torch.full((int(self._unit_vocab(u)),), t0, dtype=torch.int32, device=device)
                           for u in range(len(plan.units))]
            self._key = None
        self._ensure_adam_tables(device)
        plan.ensure_gacc()
        key = [str(device)]
        for p in plan.table_params:
            key.append((p.data_ptr(), _GACC[p].data_ptr(), p.stride(0)) + tuple(
                (d[id(p)].data_ptr(), d[id(p)].stride(0)) if id(p) in d else 0 for d in (self.s1, self.s2)))
        key = tuple(key)
        if key != self._key:
            arr = (L.LazyUnit * len(plan.units))()
            for u, (di, wi, col, _) in enumerate(plan.units):
                e = arr[u]
                fd = plan.deep[di] if di >= 0 else None
                fw = plan.wide[wi] if wi >= 0 else None
                if fd is not None:
                    p = fd.param
                    e.deep, e.deep_g = p.data_ptr(), _GACC[p].data_ptr()
                    e.deep_s1 = self.s1[id(p)].data_ptr() if id(p) in self.s1 else None
                    e.deep_s2 = self.s2[id(p)].data_ptr() if id(p) in self.s2 else None
                    e.l2_deep = self.l2.get(id(p), 0.0)
                    e.ld_deep = int(p.stride(0))
                    e.ld_deep_s1 = int(self.s1[id(p)].stride(0)) if id(p) in self.s1 else 0
                    if id(p) in self.s2 and not self.s2[id(p)].is_contiguous():
                        raise RuntimeError("the second optimizer state slab of a table must be contiguous")
                if fw is not None:
                    p = fw.param
                    e.wide, e.wide_g = p.data_ptr(), _GACC[p].data_ptr()
                    e.wide_s1 = self.s1[id(p)].data_ptr() if id(p) in self.s1 else None
                    e.wide_s2 = self.s2[id(p)].data_ptr() if id(p) in self.s2 else None
                    e.l2_wide = self.l2.get(id(p), 0.0)
                    e.ld_wide = int(p.stride(0))
                    e.ld_wide_s1 = int(self.s1[id(p)].stride(0)) if id(p) in self.s1 else 0
                e.stamp = self.stamps[u].data_ptr()
                e.vocab = self._unit_vocab(u)
                e.dim = fd.dim if fd is not None else 1
                e.col = col
            self._units_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            self._key = key

    def _unit_vocab(self, u):
        di, wi, _, _ = self.plan.units[u]
        return self.plan.deep[di].vocab if di >= 0 else self.plan.wide[wi].vocab

    def _call(self, fn, name, *mid, **kw):
        dev = self.step.device
        tail = kw.get("tail", ())
        L.check(fn(ctypes.c_void_p(self._units_dev.data_ptr()), len(self.plan.units), *mid,
                   ctypes.c_void_p(self.step.data_ptr()), ctypes.byref(self.opt), self.vec, self.max_dim,
                   *(tuple(tail) + (L.stream_handle(dev),))), name)

    def catchup(self, X, sweep=True):
        """Before the gather of a train step: bring the batch's rows to the current step.  Returns ids_t.  ``sweep=False``: a
        second catch-up of the same step (the data-parallel trainer's global batch): the step's window is swept already."""
        plan = self.plan
        self._ensure(X.device)
        B = X.shape[0]
        ids_t = torch.empty((len(plan.units), B), dtype=torch.int32, device=X.device)
        L.check(L.lib().dctr_embed_ids(None, plan.units_ptr(), len(plan.units), ctypes.c_void_p(X.data_ptr()),
                                       X.stride(0), B, ctypes.c_void_p(ids_t.data_ptr()), None,
                                       L.stream_handle(X.device)), "dctr_embed_ids")
        self._join_sweep()
        # without the sweep rows sleep geometrically long: scratch for the entries' order by gap (rows that slept equally
        # long are then replayed side by side); with it no gap exceeds K and the ordering pass costs more than it saves
        order = torch.empty((len(plan.units), B), dtype=torch.int32, device=X.device) \
            if (self.sweep_k <= 0 and self.replays) else None
        self._call(L.lib().dctr_lazy_catchup, "dctr_lazy_catchup", ctypes.c_void_p(ids_t.data_ptr()), B,
                   tail=(ctypes.c_void_p(order.data_ptr()) if order is not None else None,))
        if sweep and self.sweep_k > 0 and self.replays:
            self._fork_sweep(X.device)
        return ids_t

    def _fork_sweep(self, device):
        """The sweep of this step's window, BESIDE the rest of the step: it starts behind the catch-up (the batch's rows then
        carry the current stamp, and the sweep leaves rows at the current stamp alone), runs on its own stream while the
        gather, the tower, the update and the data-gradient step run on the caller's, and is joined before the step counter
        moves (apply) -- nothing in between writes a row the sweep writes: those passes touch the batch's rows only.
        DCTR_LAZY_SWEEP_ASYNC=0 (and CPU stand-in runs): in line."""
        dev = torch.device(device)
        if dev.type != "cuda" or os.environ.get("DCTR_LAZY_SWEEP_ASYNC", "1") == "0":
            self._call(L.lib().dctr_lazy_sweep, "dctr_lazy_sweep", int(self.plan.max_vocab), self.sweep_k)
            return
        side = _streams.side_stream(dev, "sweep")
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._call(L.lib().dctr_lazy_sweep, "dctr_lazy_sweep", int(self.plan.max_vocab), self.sweep_k)
        self._sweep_side = side

    def _join_sweep(self):
        side, self._sweep_side = self._sweep_side, None
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)

    def apply(self, ids_t):
        """After dctr_embed_update(ACCUM): step t+1 on the batch's rows, then t += 1."""
        self._ensure(ids_t.device)
        self._call(L.lib().dctr_lazy_apply, "dctr_lazy_apply", ctypes.c_void_p(ids_t.data_ptr()), ids_t.shape[1])
        self._join_sweep()          # (the sweep reads the step counter: it must be done before the counter moves)
        L.check(L.lib().dctr_lazy_step_inc(ctypes.c_void_p(self.step.data_ptr()), L.stream_handle(ids_t.device)),
                "dctr_lazy_step_inc")
        self.mark_dirty()

    def update_fused(self, plan, cplan, ids_t, parts_t, B, g_out, ld_g, out, fm_s, g_fm, g_wide, X, g_wd, ws, ws_n):
        """The data-gradient step INSIDE the sorted update (dctr_embed_update_lazy, round 6): what dctr_embed_update(ACCUM) +
        apply() did in two passes over the batch's rows through a gradient slab.  Needs pre-sorted entries (the segment
        pre-pass ran on these ids).  False: not taken (the caller runs the two passes)."""
        if os.environ.get("DCTR_LAZY_FUSED_APPLY", "1") == "0":
            return False
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())       # noqa: E731
        rc = L.lib().dctr_embed_update_lazy(
            cplan, plan.units_ptr(), plan.n_grid_units, plan.max_vocab, P(ids_t), P(parts_t), B, P(g_out), ld_g, P(out),
            plan.ld_out, P(fm_s), fm_s.stride(0) if fm_s is not None else 0, P(g_fm), P(g_wide), 1, P(X), X.stride(0), P(g_wd),
            P(ws), ws_n, ctypes.c_void_p(self._units_dev.data_ptr()), ctypes.c_void_p(self.step.data_ptr()),
            ctypes.byref(self.opt), L.stream_handle(X.device))
        if rc == L.ENOSUP:
            return False
        L.check(rc, "dctr_embed_update_lazy")
        self._join_sweep()          # (the sweep reads the step counter: it must be done before the counter moves)
        L.check(L.lib().dctr_lazy_step_inc(ctypes.c_void_p(self.step.data_ptr()), L.stream_handle(X.device)),
                "dctr_lazy_step_inc")
        self.mark_dirty()
        return True

    def mark_dirty(self):
        """Train steps ran since the last flush (LazyState.apply, or a hipGraph replay of it): rows lag behind."""
        self.dirty = True
        self._reg = None

    def flush(self, device=None):
        """Bring EVERY row to the current step (before predict / evaluate / state_dict read the tables)."""
        self._join_sweep()
        if not self.dirty or self.step is None:
            return
        self._ensure(self.step.device)
        self._call(L.lib().dctr_lazy_flush, "dctr_lazy_flush", int(self.plan.max_vocab))
        self.dirty = False
        if self.optimizer is not None and self.kind in ("adam", "rmsprop") and not (
                self.step.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            t = float(int(self.step.item()))
            for p in self.plan.table_params:
                st = self.optimizer.state.get(p)
                if st is not None and "step" in st:
                    st["step"] = torch.tensor(t, dtype=st["step"].dtype) if torch.is_tensor(st["step"]) else t

    def reg_value(self, device):
        """lambda * sum(w^2) over the lazily regularised tables (the term get_regularization_loss adds to the LOGGED
        loss; its gradient is applied by the kernels).  Small models (<= EXACT_REG_ELEMS table elements): flushed and
        summed at every call, i.e. exact at every step.  Bigger ones: the value of the last flush -- an O(vocabulary)
        reduction per step is exactly what this mode exists to avoid."""
        if not any(v > 0 for v in self.l2.values()):
            return None
        exact = self.n_elems <= self.EXACT_REG_ELEMS
        if exact:
            self.flush()
        if self._reg is None and (exact or not self.dirty or self._last_reg is None):
            tot = torch.zeros((1,), device=device)
            for p in self.plan.table_params:
                lam = self.l2.get(id(p), 0.0)
                if lam > 0:
                    tot = tot + torch.sum(lam * torch.square(p.detach()))
            self._reg = self._last_reg = tot
        return self._reg if self._reg is not None else self._last_reg


class EmbeddingPlan(object):
    """Host mirror of ``dctr_plan_t`` + the device arrays it points to.

    deep side  = ``dnn_feature_columns`` over ``embedding_dict``          (basemodel.py:354-380)
    wide side  = ``linear_feature_columns`` over ``Linear.embedding_dict`` (basemodel.py:63-92)
    """

    def __init__(self, feature_index, deep_columns=(), deep_tables=None, wide_columns=(), wide_tables=None,
                 wide_dense_weight=None, unpooled=False, with_dense=True):
        self.feature_index = feature_index
        self.n_xcols = max([hi for (_, hi) in feature_index.values()] + [1])
        dfix, dpool, emb_width = _fields_for(deep_columns, deep_tables, feature_index, unpooled) \
            if deep_tables is not None else ([], [], 0)
        wfix, wpool, _ = _fields_for(wide_columns, wide_tables, feature_index, False) \
            if wide_tables is not None else ([], [], 0)
        self.deep = dfix + dpool
        self.wide = wfix + wpool
        self.n_deep_fixed, self.n_wide_fixed = len(dfix), len(wfix)
        self.emb_width = emb_width
        dims = sorted(set(f.dim for f in self.deep))
        self.emb_dim = dims[0] if len(dims) == 1 else 0
        self.max_dim = max(dims) if dims else 0
        self.vec = 4 if all(d % 4 == 0 for d in dims) else (2 if all(d % 2 == 0 for d in dims) else 1)
        self.has_maxpool = any(f.pool == 3 for f in self.deep + self.wide)

        _, _, ddense = split_columns(deep_columns)
        _, _, wdense = split_columns(wide_columns)
        self.dense_cols = [c for fc in ddense for c in range(*feature_index[fc.name])] if with_dense else []
        self.wdense_cols = [c for fc in wdense for c in range(*feature_index[fc.name])]
        self.wide_dense_weight = wide_dense_weight if self.wdense_cols else None
        self.dense_off = emb_width if self.dense_cols else -1
        self.width = emb_width + len(self.dense_cols)      # logical row width (combined_dnn_input)
        self.ld_out = max(4, (self.width + 3) // 4 * 4)     # padded leading dimension of `out`

        # units of the deterministic update kernel (csrc/update_kernels.hpp).  SIMPLE plans (every field fixed-length, no
        # table shared): a unit is one id column of X with the deep and / or wide table it feeds -- the layout the kernels
        # have always run, also what the lazy update, the sharded trainers and the two-level pre-pass assume.  Otherwise
        # (pooled VarLen fields, tables shared through embedding_name): GENERAL units, see _build_general_units.
        all_fields = self.deep + self.wide
        self.max_vocab = max([f.vocab for f in all_fields] + [1])
        self.simple_units = (len(all_fields) > 0 and all(f.len == 1 and f.pool == 0 for f in all_fields) and
                             len(set(id(f.param) for f in all_fields)) == len(all_fields))
        self.units = []
        self.gen = None          # host image of dctr_plan_ext_t (general units), None for simple plans
        if self.simple_units or not all_fields:
            used_wide = set()
            for i, f in enumerate(self.deep):
                j = next((k for k, w in enumerate(self.wide) if k not in used_wide and w.col == f.col and
                          w.vocab == f.vocab and w.len == 1), -1)
                if j >= 0:
                    used_wide.add(j)
                self.units.append((i, j, f.col, 0))
            for k, w in enumerate(self.wide):
                if k not in used_wide:
                    self.units.append((-1, k, w.col, 0))
            self.unit_path = self.simple_units
        else:
            self.unit_path = self._build_general_units()

        self._params = []  # unique table parameters, first-use order
        seen = set()
        for f in self.deep + self.wide:
            if id(f.param) not in seen:
                seen.add(id(f.param))
                self._params.append(f.param)
        self.version = 0   # bumps whenever device pointers were re-baked (HIP graphs must re-capture)
        self._owner = None
        self._update = ("dense",)
        self.exchange = None   # set by parallel.DataParallelTrainer: backward hands row gradients over
        self.sharder = None    # set by parallel.ShardedTrainer: lookups go through the table-sharded exchange
        self.dense_sink = None
        self._lazy = None      # LazyState when the tables take the exact lazy regularised / Adam update
        self._reset_device_image()

    # ---- general update units (include/dctr.h: dctr_plan_ext_t) -----------------------------------------------------
    def _build_general_units(self):
        """A unit = (deep table | none, wide table | none) + every X column that feeds it: the positions of a pooled
        VarLenSparseFeat (inputs.py:141-155) and every column that shares the table through ``embedding_name``
        (inputs.py:158-180).  A deep group (all deep fields over one table) pairs with a wide group when their columns
        match one to one -- same X column, pooling, length source -- as they do when ``linear_feature_columns`` and
        ``dnn_feature_columns`` name the same features; otherwise each side is a unit of its own.  Returns False when the
        kernels' envelope does not hold it (then the atomic two-pass path of csrc/embed.hip runs)."""
        def groups(fields):
            g = {}
            for i, f in enumerate(fields):
                g.setdefault(id(f.param), []).append(i)
            return list(g.values())

        def sigs(fields, idxs):
            return [((f.col + t, f.pool, f.len, f.len_col, t), i) for i in idxs for f in (fields[i],) for t in range(f.len)]

        dgroups, wgroups = groups(self.deep), groups(self.wide)
        wsig = [sigs(self.wide, g) for g in wgroups]
        used = set()
        units = []          # (deep slots [(sig, di)], wide slots aligned with them | None)
        for g in dgroups:
            ds = sigs(self.deep, g)
            vocab = self.deep[g[0]].vocab
            hit = None
            for k, ws in enumerate(wsig):
                if k in used or self.wide[wgroups[k][0]].vocab != vocab or len(ws) != len(ds):
                    continue
                if sorted(x[0] for x in ws) == sorted(x[0] for x in ds) and len(set(x[0] for x in ds)) == len(ds):
                    hit = k
                    break
            if hit is None:
                units.append((ds, None))
            else:
                used.add(hit)
                by_sig = dict((x[0], x[1]) for x in wsig[hit])
                units.append((ds, [(x[0], by_sig[x[0]]) for x in ds]))
        for k, ws in enumerate(wsig):
            if k not in used:
                units.append((None, ws))

        am_deep, am_wide, ld_am = [-1] * len(self.deep), [-1] * len(self.wide), 0
        for i, f in enumerate(self.deep):
            if f.pool == 3:
                am_deep[i], ld_am = ld_am, ld_am + f.dim
        for i, f in enumerate(self.wide):
            if f.pool == 3:
                am_wide[i], ld_am = ld_am, ld_am + 1
        ld_am = (ld_am + 15) // 16 * 16
        slots, vunits, vocabs, den_rows = [], [], [], {}
        self.units = []
        for ds, ws in units:
            ref = ds if ds is not None else ws
            ns = len(ref)
            if ns > L.MAX_UNIT_SLOTS:
                return False
            c0, vu0 = len(slots), len(vunits)
            di = ds[0][1] if ds is not None else -1
            wi = ws[0][1] if ws is not None else -1
            for n, (sig, _) in enumerate(ref):
                col, pool, length, len_col, t = sig
                if pool == 3 and length > 255:
                    return False
                fdi = ds[n][1] if ds is not None else -1
                fwi = ws[n][1] if ws is not None else -1
                den = -1
                if pool == 2:
                    den = den_rows.setdefault((col - t, length, len_col), len(den_rows))
                slots.append(dict(col=col, goff=self.deep[fdi].out_off if fdi >= 0 else -1, wide=1 if fwi >= 0 else 0,
                                  pool=pool, t=t, len=length, len_col=len_col, den=den,
                                  am_deep=am_deep[fdi] if fdi >= 0 else -1, am_wide=am_wide[fwi] if fwi >= 0 else -1,
                                  vu0=vu0))
            k = ns                      # groups of P partitions: a partition then holds ~96 entries whatever the slots
            kshift = 32 + max(0, (k - 1).bit_length())
            for j in range(k):
                vunits.append(dict(di=di, wi=wi, c0=c0, n_slots=ns, k=k, j=j, kshift=kshift,
                                   kmagic=((1 << kshift) // k + 1) & 0xFFFFFFFFFFFFFFFF))
                vocabs.append(self.deep[di].vocab if di >= 0 else self.wide[wi].vocab)
            self.units.append((di, wi, ref[0][0][0], 0))
        self.gen = dict(slots=slots, vunits=vunits, vocabs=vocabs, am_deep=am_deep, am_wide=am_wide, ld_amax=ld_am,
                        n_den=len(den_rows), max_unit_slots=max(len(d if d is not None else w) for d, w in units))
        return True

    @property
    def n_vcols(self):
        """Rows of ids_t / parts_t: one per X column feeding a unit (= len(units) for simple plans)."""
        return len(self.gen["slots"]) if self.gen is not None else len(self.units)

    @property
    def n_grid_units(self):
        """What the update entry points take as ``n_units``: the vunits of a general plan, else the units."""
        return len(self.gen["vunits"]) if self.gen is not None else len(self.units)

    def step_buffers(self, B, device):
        """(den_t [n_den, B] | None, amax [B, ld_amax] u8 | None): per-step side buffers of a general plan."""
        if self.gen is None:
            return None, None
        den = torch.empty((self.gen["n_den"], B), dtype=torch.float32, device=device) if self.gen["n_den"] else None
        am = torch.empty((B, self.gen["ld_amax"]), dtype=torch.uint8, device=device) if self.gen["ld_amax"] else None
        return den, am

    def point_step_buffers(self, den_t, amax):
        """Aim the ext block at this step's side buffers (read at enqueue time, like dctr_plan_t.step_sync)."""
        self._step_bufs = (den_t, amax)         # (bind() re-creates the ext block when a table pointer moved)
        if self.gen is not None and self.cext is not None:
            self.cext.den_t = den_t.data_ptr() if den_t is not None else None
            self.cext.amax = amax.data_ptr() if amax is not None else None

    @property
    def lazy(self):
        return self._owner.lazy if self._owner is not None else self._lazy

    @lazy.setter
    def lazy(self, value):
        self._lazy = value

    def _reset_device_image(self):
        self._key = None
        self._dev = {}
        self.cplan = L.Plan()
        self.cext = None
        self._host_ext = None
        self._step_bufs = (None, None)
        self.anchor = None
        self._err = None
        self._wd_idx = None
        self._upd_ws = {}
        self._seg_stream = None

    # models holding a plan stay picklable (tests/utils.py:162-170 of the reference pickle whole models):
    # raw ctypes / device handles are dropped and re-baked lazily
    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ("_key", "_dev", "cplan", "cext", "_host_ext", "_step_bufs", "anchor", "_err", "_wd_idx", "_upd_ws", "_seg_stream"):
            d.pop(k, None)
        d["exchange"] = None
        d["sharder"] = None
        d["dense_sink"] = None
        d["_lazy"] = None          # re-created by the model's compile() / first train step
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._reset_device_image()

    # ---- sparse-update mode (see ops.py) ---------------------------------------------------------
    @property
    def update(self):
        return self._owner.update if self._owner is not None else self._update

    @update.setter
    def update(self, value):
        self._update = tuple(value)

    def share_update_with(self, owner):
        """Secondary plans over the same tables follow the owner plan's sparse-update mode."""
        self._owner = owner

    # ---- slabs ---------------------------------------------------------------------------------
    @property
    def table_params(self):
        return list(self._params)

    def ensure_gacc(self):
        """Allocate the zero-at-rest gradient slab of every table (idempotent)."""
        for p in self._params:
            g = _GACC.get(p)
            if g is None or g.device != p.device or g.shape != p.shape:
                _GACC[p] = torch.zeros_like(p.data)
        return self

    def gacc_of(self, param):
        return _GACC.get(param)

    def set_state(self, mapping):
        """``{param: slab}`` -- optimizer state tensors (e.g. ``optimizer.state[p]['sum']``)."""
        for p in self._params:
            slab = None
            for q, s in (mapping or {}).items():
                if q is p:
                    slab = s
            if slab is not None:
                _STATE[p] = slab
            elif p in _STATE:
                del _STATE[p]

    def prepare_dense_grads(self):
        """``param.grad`` := the gacc slab (autograd semantics: the first backward after zero_grad starts
        from zero, later backwards accumulate)."""
        for p in self._params:
            slab = _GACC[p]
            if not p.requires_grad:
                # a frozen table (pretrained embeddings): like autograd, hand the optimizer no gradient.  The scatter
                # still lands in the slab; it is zeroed before it is next used as a gradient.
                slab._dctr_dirty = True
                continue
            if p.grad is None:
                if getattr(slab, "_dctr_dirty", False):
                    slab.zero_()
                p.grad = slab
            elif p.grad.data_ptr() != slab.data_ptr():
                # autograd got there first (e.g. the L2 term of get_regularization_loss deposited its dense
                # gradient before the lookup's backward ran): adopt it -- O(V), but so was that gradient
                if p.grad.shape != slab.shape or p.grad.is_sparse:
                    raise RuntimeError("embedding parameter .grad was replaced by an incompatible tensor; "
                                       "call optimizer.zero_grad() before backward")
                slab.copy_(p.grad)
                p.grad = slab
            slab._dctr_dirty = True

    def err_flag(self, device):
        if self._err is None or self._err.device != torch.device(device):
            self._err = torch.zeros(1, dtype=torch.int32, device=device)
        return self._err

    def check_ids(self):
        """Poll the out-of-range flag (one device sync).  The reference raises IndexError at once on CPU."""
        if self._err is not None:
            bits = int(self._err.item())
            if bits != 0:
                self._err.zero_()
                import os
                if (bits & 12) and os.environ.get("DCTR_DBG_IGNORE_WAIT") == "1":       # (timing experiments only)
                    import sys
                    print("dctr: an in-kernel wait timed out (bits %d) -- ignored (DCTR_DBG_IGNORE_WAIT)" % bits, file=sys.stderr)
                    return
                if bits & 12:   # (dctr_embed_tower_train_step_sync: its wait for the weights' generation ran out)
                    raise RuntimeError("a train step's tower launch gave up waiting for the previous step's dense optimizer "
                                       "step (DCTR_SYNC_W_GEN): the two queues of the step fell out of step -- results "
                                       "since the last check are not to be trusted (DCTR_STEP_TOPOLOGY=update_side avoids "
                                       "the in-kernel wait)")
                raise IndexError("index out of range in self: a sparse id in X is outside [0, vocabulary_size)")
        owner = getattr(self, "_sync_owner", None)      # step topology "flags": did a device-side dependency time out?
        if owner is not None:
            owner.check_sync()

    def dense_matrix(self, X, cols):
        lo, hi = cols[0], cols[-1] + 1
        if list(cols) == list(range(lo, hi)):
            return X[:, lo:hi]
        if self._wd_idx is None or self._wd_idx.device != X.device:
            self._wd_idx = torch.tensor(list(cols), dtype=torch.long, device=X.device)
        return X.index_select(1, self._wd_idx)

    # ---- device image --------------------------------------------------------------------------
    def _pointer_key(self, device):
        key = [str(device)]
        for p in self._params:
            g, s = _GACC.get(p), _STATE.get(p)
            key.append((p.data_ptr(), g.data_ptr() if g is not None else 0, s.data_ptr() if s is not None else 0,
                        p.stride(0), s.stride(0) if s is not None else 0))
        w = self.wide_dense_weight
        key.append(w.data_ptr() if w is not None else 0)
        return tuple(key)

    def _field_bytes(self, specs):
        arr = (L.Field * max(1, len(specs)))()
        for i, f in enumerate(specs):
            g, s = _GACC.get(f.param), _STATE.get(f.param)
            arr[i].table = f.param.data_ptr()
            arr[i].gacc = g.data_ptr() if g is not None else None
            arr[i].state = s.data_ptr() if s is not None else None
            arr[i].vocab, arr[i].dim, arr[i].col, arr[i].len = f.vocab, f.dim, f.col, f.len
            arr[i].pool, arr[i].len_col, arr[i].out_off = f.pool, f.len_col, f.out_off
            # row strides: a table / its state may be strided views of one interleaved slab (_hip/layout.py)
            arr[i].ld = int(f.param.stride(0))
            arr[i].ld_state = int(s.stride(0)) if s is not None else 0
        return bytes(arr)

    def bind(self, device):
        """Return ``ctypes.byref(dctr_plan_t)`` valid for the tables' CURRENT device pointers."""
        device = torch.device(device)
        key = self._pointer_key(device)
        if key == self._key:
            return ctypes.byref(self.cplan)
        for p in self._params:
            L.require_gpu(p, "embedding table")
            if p.dim() != 2 or p.dtype != torch.float32 or (p.shape[1] > 1 and p.stride(1) != 1) or \
                    p.stride(0) < p.shape[1]:
                raise RuntimeError("embedding tables must be float32 [V, D] with unit stride inside a row")
            if self.vec > 1 and p.shape[1] > 1 and (p.stride(0) % self.vec or p.data_ptr() % (4 * self.vec)):
                raise RuntimeError("embedding table rows must start on %d-byte boundaries" % (4 * self.vec))
            st = _STATE.get(p)
            if st is not None and (st.shape != p.shape or (p.shape[1] > 1 and st.stride(1) != 1) or
                                   (self.vec > 1 and p.shape[1] > 1 and
                                    (st.stride(0) % self.vec or st.data_ptr() % (4 * self.vec)))):
                raise RuntimeError("optimizer state of an embedding table must mirror its shape and alignment")
        if self.wide_dense_weight is not None:
            L.require_gpu(self.wide_dense_weight, "Linear.weight")

        def up(raw, dtype):
            return torch.frombuffer(bytearray(raw), dtype=dtype).to(device)

        import numpy as np
        self._dev = {
            "deep": up(self._field_bytes(self.deep), torch.uint8),
            "wide": up(self._field_bytes(self.wide), torch.uint8),
            "dense": up(np.asarray(self.dense_cols or [0], dtype=np.int32).tobytes(), torch.int32),
            "wdense": up(np.asarray(self.wdense_cols or [0], dtype=np.int32).tobytes(), torch.int32),
            "units": up(np.asarray(self.units or [(-1, -1, 0, 0)], dtype=np.int32).tobytes(), torch.int32),
        }
        c = self.cplan
        c.deep = self._dev["deep"].data_ptr() if self.deep else None
        c.wide = self._dev["wide"].data_ptr() if self.wide else None
        c.dense_cols = self._dev["dense"].data_ptr()
        c.wdense_cols = self._dev["wdense"].data_ptr()
        w = self.wide_dense_weight
        c.wdense_w = w.data_ptr() if w is not None else None
        c.n_deep, c.n_deep_fixed = len(self.deep), self.n_deep_fixed
        c.n_wide, c.n_wide_fixed = len(self.wide), self.n_wide_fixed
        c.n_dense, c.n_wdense = len(self.dense_cols), (len(self.wdense_cols) if w is not None else 0)
        c.dense_off, c.emb_dim, c.n_xcols = self.dense_off, self.emb_dim, self.n_xcols
        c.max_dim, c.vec = self.max_dim, self.vec
        flags = 0
        if self._params and all(p in _GACC for p in self._params):
            flags |= L.PLAN_HAS_GACC
        if self._params and all(p in _STATE for p in self._params):
            flags |= L.PLAN_HAS_STATE
        if self.has_maxpool:
            flags |= L.PLAN_HAS_MAXPOOL
        c.flags = flags
        c.ext = None
        if self.gen is not None:
            g = self.gen
            sl = (L.USlot * len(g["slots"]))()
            for i, d in enumerate(g["slots"]):
                for k_, v in d.items():
                    setattr(sl[i], k_, v)
            vu = (L.VUnit * len(g["vunits"]))()
            for i, d in enumerate(g["vunits"]):
                for k_, v in d.items():
                    setattr(vu[i], k_, v)
            hv = (ctypes.c_int64 * len(g["vocabs"]))(*g["vocabs"])
            self._dev["slots"] = up(bytes(sl), torch.uint8)
            self._dev["vunits"] = up(bytes(vu), torch.uint8)
            self._dev["am_deep"] = up(np.asarray(g["am_deep"] or [-1], dtype=np.int32).tobytes(), torch.int32)
            self._dev["am_wide"] = up(np.asarray(g["am_wide"] or [-1], dtype=np.int32).tobytes(), torch.int32)
            x = L.PlanExt()
            x.slots, x.vunits = self._dev["slots"].data_ptr(), self._dev["vunits"].data_ptr()
            x.am_deep_off, x.am_wide_off = self._dev["am_deep"].data_ptr(), self._dev["am_wide"].data_ptr()
            x.h_vunits, x.h_vocab = ctypes.addressof(vu), ctypes.addressof(hv)
            x.den_t = x.amax = None
            x.n_vcols, x.n_vunits, x.n_units = len(g["slots"]), len(g["vunits"]), len(self.units)
            x.max_unit_slots, x.n_den, x.ld_amax = g["max_unit_slots"], g["n_den"], g["ld_amax"]
            gsd = [(i << 16) | t for i, f in enumerate(self.deep) if f.pool != 0 for t in range(f.len)]
            gsw = [(i << 16) | t for i, f in enumerate(self.wide) if f.pool != 0 for t in range(f.len)]
            self._dev["gslot_deep"] = up(np.asarray(gsd or [0], dtype=np.int32).tobytes(), torch.int32)
            self._dev["gslot_wide"] = up(np.asarray(gsw or [0], dtype=np.int32).tobytes(), torch.int32)
            x.gslot_deep, x.gslot_wide = self._dev["gslot_deep"].data_ptr(), self._dev["gslot_wide"].data_ptr()
            x.n_gslot_deep, x.n_gslot_wide = len(gsd), len(gsw)
            self.cext, self._host_ext = x, (vu, hv)
            c.ext = ctypes.addressof(x)
            self.point_step_buffers(*getattr(self, "_step_bufs", (None, None)))
        self._key = key
        self.version += 1
        if self.anchor is None or self.anchor.device != device:
            # a leaf that requires grad: makes autograd call EmbedFunction.backward even when no
            # differentiable tensor enters the lookup (tables are updated in place, not via autograd)
            self.anchor = torch.zeros(1, device=device, requires_grad=True)
        return ctypes.byref(self.cplan)

    def units_ptr(self):
        return ctypes.c_void_p(self._dev["units"].data_ptr())

    def update_workspace(self, B, device, always=False, slot=0):
        """(int32 tensor | None, n_ints): the bucket workspace of ``dctr_embed_update`` / ``dctr_embed_segments`` -- zero
        before its first use, left ready by the kernels, so one tensor per (batch, device) serves every step.  Without
        the segment pre-pass it only pays for large batches, where a workgroup's scan over the unit's B ids is the
        expensive part (DCTR_UPD_BUCKET=1 / 0 forces it on / off); ``always``: the pre-pass needs it at any size."""
        import os
        mode = os.environ.get("DCTR_UPD_BUCKET", "auto")
        if not always and (mode == "0" or (mode != "1" and B < 8192)):
            return None, 0
        # (slot: the "tower_seg" step topology alternates between two workspaces -- the pre-pass of step n runs on the
        # side stream while the update of step n-1 may still be reading its own on the main stream)
        key = (int(B), str(device)) if not slot else (int(B), str(device), int(slot))
        ws = self._upd_ws.get(key)
        if ws is not None:
            self._upd_ws[key] = self._upd_ws.pop(key)      # (least recently used goes first)
        if ws is None:
            n = int(L.lib().dctr_embed_update_workspace_ints(ctypes.byref(self.cplan), self.n_grid_units, int(B)))
            ws = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
            # One workspace per batch size, kept: a captured hipGraph holds the raw address of the one it was captured
            # with (fit() alternates between the full-size batch's graph and an eager ragged last batch -- dropping
            # the first workspace when the second is made left the graph writing into freed memory).  When a run
            # keeps inventing batch sizes the oldest goes, and the plan version bump makes captured steps re-capture.
            if len(self._upd_ws) >= 8:
                self._upd_ws.pop(next(iter(self._upd_ws)))
                self.version += 1
            self._upd_ws[key] = ws
        return ws, ws.numel()

    # ---- the segment pre-pass (dctr_embed_segments): what the update needs of the ids, computed right after the
    # forward on a side stream, in the shadow of the tower ------------------------------------------------------
    def segments_enabled(self):
        import os
        return os.environ.get("DCTR_SEGMENTS", "1") != "0"

    def launch_segments(self, ids_t, parts_t, B, X=None, before=None, fork=True, slot=0, join_before=True):
        """Enqueue the pre-pass for this forward's ids on the side stream.  Returns the handle the update passes to
        ``update_workspace_for``.  A workspace still marked by an earlier forward (whose backward never ran -- a
        forward in train mode that was not followed by a backward) is taken over.
        ``before(stream_handle)``: work that goes to the side stream ahead of the pre-pass -- the gather itself in the
        "gather_side" step topology; the calling stream then waits for exactly that work (not for the pre-pass).
        ``fork=False``: the side stream does not wait for the calling stream first (its own order -- behind the
        previous step's update -- is all the gather needs)."""
        device = ids_t.device
        ws, ws_n = self.update_workspace(B, device, always=True, slot=slot)
        dirty = getattr(ws, "_dctr_owner", None) is not None

        def enqueue(stream):
            if dirty:
                ws.zero_()          # the abandoned pre-pass left bucket counts behind
            if X is not None:
                # the ids (and their partition tags) straight from X: the side stream's chain -- ids, pre-pass, and
                # later the update -- then hangs on nothing the gather kernel produces (one queue crossing less on
                # the step's critical chain; the gather skips these two side outputs)
                L.check(L.lib().dctr_embed_ids(ctypes.byref(self.cplan), self.units_ptr(), self.n_grid_units,
                                               ctypes.c_void_p(X.data_ptr()), X.stride(0), int(B),
                                               ctypes.c_void_p(ids_t.data_ptr()), ctypes.c_void_p(parts_t.data_ptr()),
                                               stream), "dctr_embed_ids")
            L.check(L.lib().dctr_embed_segments(ctypes.byref(self.cplan), self.units_ptr(), self.n_grid_units,
                                                self.max_vocab, ctypes.c_void_p(ids_t.data_ptr()),
                                                ctypes.c_void_p(parts_t.data_ptr()), int(B),
                                                ctypes.c_void_p(ws.data_ptr()), ws_n, stream), "dctr_embed_segments")

        # Ownership of the workspace is a monotonically increasing token kept in the handle and on the workspace -- not the
        # address of ids_t: the caching allocator hands a freed address to the next forward, and a stale backward would
        # then accept another forward's sorted buckets as its own (round-2 advisor finding).
        _OWNER_TOKEN[0] += 1
        token = _OWNER_TOKEN[0]
        if device.type != "cuda":                  # (CPU stand-in: same calls, no streams)
            if before is not None:
                before(None)
            enqueue(None)
            ws._dctr_owner = token
            return (True, ws, token)
        main = torch.cuda.current_stream(device)
        side = self._seg_stream
        if side is None or side.device != device:
            side = self._seg_stream = _streams.side_stream(device, "seg")
        if fork:
            side.wait_stream(main)
        if before is not None:
            with torch.cuda.stream(side):
                before(L.stream_handle(device))
            if join_before:
                main.wait_stream(side)      # (an event at the side stream's tail of NOW: the pre-pass comes behind it)
            # (join_before=False: the caller orders the main stream behind `before`'s work itself -- dctr_step_wait)
        with torch.cuda.stream(side):
            enqueue(L.stream_handle(device))
        ws._dctr_owner = token
        # (the consumer joins with wait_stream, whose events come from torch's pool: a torch.cuda.Event created here
        # would be destroyed by the garbage collector at some later point -- possibly while a hipGraph capture is
        # running, which HIP answers with hipErrorStreamCaptureUnsupported from inside a destructor: abort)
        return (side, ws, token)

    def update_workspace_for(self, ids_t, handle, B):
        """(workspace | None, n_ints, presorted) for the update of the forward that produced ``ids_t``: its own
        pre-pass when that ran (after waiting for it); otherwise the plain bucket workspace of large batches -- unless
        another forward's pre-pass currently owns it, then none (every workgroup scans for itself)."""
        device = ids_t.device
        if handle is not None:
            event, ws, token = handle
            if getattr(ws, "_dctr_owner", None) == token:
                ws._dctr_owner = None
                if event is not True and torch.cuda.current_stream(device) != event:
                    # the side stream: the pre-pass is its last work.  (When the update itself runs on that stream --
                    # the fused step with in-kernel optimizer -- stream order is all it takes; a stream waiting for
                    # itself inside a hipGraph capture crashed hipStreamEndCapture.)
                    torch.cuda.current_stream(device).wait_stream(event)
                return ws, ws.numel(), 1
        ws, n = self.update_workspace(B, device)
        if ws is not None and getattr(ws, "_dctr_owner", None) is not None:
            return None, 0, 0
        return ws, n, 0

    def update_kernel_ok(self, B):
        """True when the deterministic fused update (dctr_embed_update) can run this plan at batch ``B``."""
        if not self.unit_path or B <= 0:
            return False
        return bool(L.lib().dctr_embed_update_supported(ctypes.byref(self.cplan), self.max_vocab, int(B)))

    @property
    def has_lookup(self):
        return bool(self.deep or self.dense_cols)

    @property
    def has_wide(self):
        return bool(self.wide or (self.wdense_cols and self.wide_dense_weight is not None))


_OWNER_TOKEN = [0]       # see EmbeddingPlan.launch_segments

__all__ = ["EmbeddingPlan", "DenseFeat", "SparseFeat", "VarLenSparseFeat"]
