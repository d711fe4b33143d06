# -*- coding: utf-8 -*-
"""Data-parallel training: one process per GPU, RCCL over xGMI (``torch.distributed`` backend ``"nccl"``).

The reference's only multi-GPU path is ``torch.nn.DataParallel`` inside ONE process
(``basemodel.py:206-209``): every step it re-broadcasts all parameters (1.77 GB at the Criteo shape), scatters
the batch, gathers outputs and reduce-adds dense ``[V, D]`` gradients onto ``gpus[0]``.  Here each rank keeps a
replica and trains on its own shard of the global batch; per step there are exactly two exchanges:

  dense parameters (0.57 MB for DeepFM)   one flat bucket, ``all_reduce(SUM)`` -- SUM, not mean, because the
                                          reference's loss is ``reduction='sum'`` over the global batch
                                          (``basemodel.py:209,254``);
  embedding rows                          each rank folds FM's backward into per-sample row gradients and
                                          all-gathers one packed payload ``[B, G | g_wide | ids]``; every
                                          replica then runs the SAME deterministic fused update
                                          (``dctr_embed_update``, csrc/update.hip) over the global batch.
                                          The kernel has no atomics and sums duplicate ids in (id, sample)
                                          order, so replicas stay bit-identical without any parameter
                                          broadcast -- which is why float atomics were not an option here.

Semantics = the reference's: one optimizer step on the gradient summed over ``world_size x batch`` samples.

The exchange logic (payload layout, bucket views, collectives) is plain torch and is exercised on CPU with the
``gloo`` backend in ``tests/test_parallel_gloo.py``; only the kernels need a GPU.
"""
import os

import torch
import torch.distributed as dist


def _broadcast(t, src, group):
    """dist.broadcast that also serves strided views (tables seated in an interleaved slab, _hip/layout.py):
    collectives want dense buffers."""
    if t.is_contiguous():
        dist.broadcast(t, src, group=group)
    else:
        buf = t.contiguous()
        dist.broadcast(buf, src, group=group)
        t.copy_(buf)


class DenseBucket(object):
    """All dense (non-table) gradients live in ONE flat fp32 buffer; ``param.grad`` are views into it, so the
    all-reduce needs no packing copies and is a single collective (the whole DeepFM tower is 0.57 MB:
    latency-bound, so one message beats per-tensor messages)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def attach(self):
        """Zero the bucket and point every ``param.grad`` at its view (call between forward and backward)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, group=None, async_op=False):
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class SparsePayload(object):
    """Layout of the per-sample row every rank contributes to the embedding exchange:

        [ G : g_width floats | g_wide : 1 | pad to 4 | X : n_xcols floats | pad to 4 ]

    ``G`` first so that its base pointer and the row stride stay 16-byte aligned for float4 access."""

    def __init__(self, g_width, n_xcols):
        self.g_width = int(g_width)
        self.n_xcols = int(n_xcols)
        self.off_gw = self.g_width
        self.off_x = (self.g_width + 1 + 3) // 4 * 4
        self.ld = (self.off_x + self.n_xcols + 3) // 4 * 4

    def pack(self, X, G, g_wide):
        B = X.shape[0]
        row = torch.zeros((B, self.ld), dtype=torch.float32, device=X.device)
        if G is not None and self.g_width:
            row[:, :self.g_width] = G[:, :self.g_width]
        if g_wide is not None:
            row[:, self.off_gw] = g_wide
        row[:, self.off_x:self.off_x + self.n_xcols] = X[:, :self.n_xcols]
        return row

    def gather(self, row, group=None):
        world = dist.get_world_size(group)
        out = torch.empty((world * row.shape[0], self.ld), dtype=row.dtype, device=row.device)
        dist.all_gather_into_tensor(out, row, group=group)
        return out

    def views(self, gathered):
        """(X_all [NB, n_xcols] view, G_all [NB, g_width] view, g_wide_all [NB] contiguous)"""
        return (gathered[:, self.off_x:self.off_x + self.n_xcols], gathered[:, :self.g_width],
                gathered[:, self.off_gw].contiguous())


def fold_fm(g_out, width, out, fm_s, g_fm, emb_dim):
    """Row gradients with FM's backward folded in (interaction.py:26-34 under autograd):
    ``G[b, f, :] = g_out[b, f, :] + g_fm[b] * (S[b, :] - e[b, f, :])``.  Plain torch, any device."""
    B = out.shape[0] if out is not None else g_out.shape[0]
    G = g_out[:, :width] if g_out is not None else None
    if g_fm is not None:
        nf = width // emb_dim
        e = out[:, :width].reshape(B, nf, emb_dim)
        fold = (g_fm.reshape(B, 1, 1) * (fm_s[:, :emb_dim].unsqueeze(1) - e)).reshape(B, width)
        G = fold if G is None else G + fold
    return G


class DataParallelTrainer(object):
    """``trainer.train_step(xb, yb)`` == ``model._train_step`` on the concatenation of every rank's batch."""

    def __init__(self, model, process_group=None, broadcast_parameters=True):
        if not dist.is_initialized():
            raise RuntimeError("initialise torch.distributed first (backend 'nccl' = RCCL on ROCm)")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.plan = model.model_plan()
        self._forced_dense = False
        # The lazy regularised / Adam table update (the reference's DEFAULT kwargs: L2 on every table, adam) stays lazy on
        # replicated tables (round 6): a row's replayed steps are a function of (row, step) alone, so every replica that
        # catches a row up -- its own batch's rows before its gather, the other ranks' rows before the global data-gradient
        # step -- computes the same bits, every replica applies the same global gradient sums in the same order, and the
        # replicas' tables, moments, stamps and step counters stay identical.  One train step is then O(global batch) like the
        # single-GPU one (0.4 ms instead of the 14 ms of the dense route at the Criteo shape).  DCTR_DP_LAZY=0: the exact
        # dense-gradient route of rounds 2-5 (the reference's own O(vocabulary) update on every replica).
        self._lazy = self.plan.update[0] == "lazy" and os.environ.get("DCTR_DP_LAZY", "1") != "0"
        if self.plan.update[0] == "lazy" and not self._lazy:
            model._no_lazy_update = True
            model._rederive_update_paths()
            self._forced_dense = True
            if self.plan.update[0] == "lazy":
                raise RuntimeError("the model kept the lazy table update although the dense route was requested")
        if not self.plan.unit_path:
            raise NotImplementedError("data-parallel training needs the deterministic update kernel; this model's "
                                      "feature columns are outside its envelope (a table fed by more than 128 columns)")
        tables = set(id(p) for p in self.plan.table_params)
        self.bucket = DenseBucket([p for p in model.parameters() if id(p) not in tables])
        self.payload = SparsePayload(self.plan.emb_width, self.plan.n_xcols)
        if broadcast_parameters:
            with torch.no_grad():
                for p in model.parameters():
                    _broadcast(p.data, 0, process_group)
                for b in model.buffers():
                    _broadcast(b.data, 0, process_group)
        self._stash = None
        self.plan.exchange = self._defer          # EmbedFunction.backward hands its inputs over instead of updating

    def close(self):
        self.plan.exchange = None
        if self._forced_dense:       # hand the model back as it was: single-GPU fit() takes the lazy O(batch) update again
            self.model._no_lazy_update = False
            self.model._rederive_update_paths()
            self._forced_dense = False

    def _regularization_terms(self):
        """get_regularization_loss() (basemodel.py:412-428 of the reference: sum(l1 |p|) + sum(l2 p^2) per registered
        weight) split into the terms on bucket parameters and the terms on embedding tables."""
        model = self.model
        tables = set(id(p) for p in self.plan.table_params)
        dense = torch.zeros((1,), device=model.device)
        tab = torch.zeros((1,), device=model.device)
        for weight_list, l1, l2 in model.regularization_weight:
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                term = None
                if l1 > 0:
                    term = torch.sum(l1 * torch.abs(p))
                if l2 > 0:
                    t2 = torch.sum(l2 * torch.square(p))
                    term = t2 if term is None else term + t2
                if term is None:
                    continue
                if id(p) in tables:
                    tab = tab + term
                else:
                    dense = dense + term
        return dense, tab

    def _regularization_terms_lazy(self):
        """The same split when the tables are on the lazy update: their term is a VALUE (LazyState.reg_value; its gradient
        is applied by the kernels), only the dense parameters' terms are autograd nodes."""
        model = self.model
        tables = set(id(p) for p in self.plan.table_params)
        dense = torch.zeros((1,), device=model.device)
        for weight_list, l1, l2 in model.regularization_weight:
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                if id(p) in tables:
                    continue
                if l1 > 0:
                    dense = dense + torch.sum(l1 * torch.abs(p))
                if l2 > 0:
                    dense = dense + torch.sum(l2 * torch.square(p))
        rv = self.plan.lazy.reg_value(model.device)
        tab = rv.detach() if rv is not None else torch.zeros((1,), device=model.device)
        return dense, tab

    def _defer(self, **kw):
        if self._stash is not None:
            raise RuntimeError("two embedding backward passes in one data-parallel step are not supported")
        self._stash = kw

    def train_step(self, xb, yb):
        from ._hip import lib as L
        from ._hip.ops import _ptr
        model, plan = self.model, self.plan
        y_pred = model(xb).squeeze()
        model.optim.zero_grad()
        self.bucket.attach()
        if isinstance(model.loss_func, list):
            loss = sum([model.loss_func[i](y_pred[:, i], yb[:, i], reduction='sum') for i in range(model.num_tasks)])
        else:
            loss = model.loss_func(y_pred, yb.squeeze(), reduction='sum')
        # Every rank holds the full dense parameters, so the regularisation / auxiliary terms -- functions of the
        # parameters, not of the samples -- would enter the SUM all-reduce of the gradients world_size times: each rank
        # contributes 1 / world of them (round-1 advisor finding).  The value that is logged is the full one.
        # ... but ONLY the terms on parameters of the all-reduced dense bucket.  Tables are not in the bucket: a table's
        # L1 / L2 gradient is applied locally by every replica's own optimizer step (dense-gradient route), unsummed, so
        # it enters whole (round-2 advisor finding: scaled by 1 / world like the rest it was under-applied world times).
        lazy = self.plan.lazy if (self._lazy and self.plan.update[0] == "lazy") else None
        reg_dense, reg_tables = self._regularization_terms() if lazy is None else self._regularization_terms_lazy()
        reg = reg_dense + reg_tables + model.aux_loss
        total_loss = loss + reg
        self._stash = None
        if lazy is None:
            (loss + (reg_dense + model.aux_loss) * (1.0 / self.world) + reg_tables).backward()
        else:       # (the tables' L2 gradient is the kernels' business)
            (loss + (reg_dense + model.aux_loss) * (1.0 / self.world)).backward()
        work = self.bucket.all_reduce(self.group, async_op=True)      # overlaps with the embedding exchange

        st = self._stash
        self._stash = None
        if st is not None:
            X = st["X"]
            G = fold_fm(st["g_out"], plan.emb_width, st["out"], st["fm_s"], st["g_fm"], plan.emb_dim) \
                if plan.deep else None
            with torch.no_grad():      # (plain data movement; a pooled field's folded gradient can arrive attached to a graph)
                gathered = self.payload.gather(self.payload.pack(X, G, st["g_wide"]).detach(), self.group)
            X_all, G_all, gw_all = self.payload.views(gathered)
            NB = gathered.shape[0]
            lib = L.lib()
            stream = L.stream_handle(X.device)
            kind = plan.update[0]
            if kind == "lazy":
                if lazy is None:
                    raise RuntimeError("the model switched to the lazy table update under a DataParallelTrainer built without it")
                # the rows of EVERY rank's samples to the current step (this rank's own were caught up before its gather:
                # their stamps say so), then the global gradient sums into the gradient slabs and ONE regularised optimizer
                # step on the touched rows (LazyState.apply: also moves the step counter)
                lazy.catchup(X_all, sweep=False)
                plan.ensure_gacc()
                opt, lr, eps = L.UPD_ACCUM, 0.0, 0.0
            elif kind == "dense":
                plan.ensure_gacc()
                plan.prepare_dense_grads()
                opt, lr, eps = L.UPD_ACCUM, 0.0, 0.0
            elif kind in ("sgd", "sgd2"):
                opt, lr, eps = L.UPD_SGD, float(plan.update[1]), 0.0
            else:
                opt, lr, eps = L.UPD_ADAGRAD, float(plan.update[1]), float(plan.update[2])
            cplan = plan.bind(X.device)
            if not plan.update_kernel_ok(NB):
                raise RuntimeError("global batch %d is beyond the deterministic update kernel" % NB)
            ids_t = torch.empty((plan.n_vcols, NB), dtype=torch.int32, device=X.device)
            parts_t = torch.empty((plan.n_vcols, NB), dtype=torch.int16, device=X.device)
            # general units (pooled VarLen fields, shared tables; round 5): mean pooling's divisors are recomputed from the
            # gathered X with the ids; max pooling's arg-max positions (a side output of every rank's own gather) are
            # all-gathered like the gradients.  FM's backward is already folded into G on the POOLED values (fold_fm).
            den_t, amax_all = plan.step_buffers(NB, X.device)
            if amax_all is not None:
                dist.all_gather_into_tensor(amax_all, st["amax"].contiguous(), group=self.group)
            plan.point_step_buffers(den_t, amax_all)
            L.check(lib.dctr_embed_ids(cplan, plan.units_ptr(), plan.n_grid_units, _ptr(X_all), gathered.stride(0), NB,
                                       _ptr(ids_t), _ptr(parts_t), stream), "dctr_embed_ids")
            L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), plan.n_grid_units, plan.max_vocab, _ptr(ids_t),
                                          _ptr(parts_t), NB,
                                          _ptr(G_all) if plan.deep else None, gathered.stride(0), None, 0, None, 0,
                                          None, _ptr(gw_all) if plan.wide else None, 1, opt, lr, eps, None, 0, None, None, None, 0, 0, stream),
                    "dctr_embed_update(global)")
            if kind == "lazy":
                lazy.apply(ids_t)
        work.wait()
        model.optim.step()
        return loss.detach(), total_loss.detach(), y_pred.detach()


# =====================================================================================================================
# Table-sharded training (the scalable path): embedding tables sharded by table across the GPUs of one node, the
# dense tower data-parallel.  SURVEY.md 8(e) option S.
# =====================================================================================================================
class ShardLayout(object):
    """Who owns what, and how a chunk row looks.

    The unit of ownership is a TABLE GROUP: one deep table with every feature column that feeds it -- a fixed-length
    SparseFeat, the positions of a pooled VarLenSparseFeat (inputs.py:141-155), columns that share the table through
    ``embedding_name`` (inputs.py:158-180) -- plus the wide tables of the same features.  Group g is owned by rank
    ``group_owner[g]`` (``assign``: balanced by slot count, then by table bytes).  Every deep FIELD f of the model (a row of
    ``plan.deep``: what ``combined_dnn_input`` concatenates) has a slot in its owner's chunk: the owner looks the field up --
    pooling a VarLen field's positions locally -- and ships ONE row of D floats per sample, whatever the field's length.  A
    chunk row is
        ``[slot 0 | slot 1 | ... | pad to 4 | wide, pad to 4 | n_ids id columns of the NEXT batch | pad to 4]``
    floats; every rank uses ``n_slots`` = the largest number of fields any rank owns and ``n_ids`` = the largest number of X
    columns any owner needs (every position of its pooled fields, their length columns), so that all exchanges have equal
    splits.  The id columns ride along with the row gradients: the ids the owners need for step k+1 travel in step k's
    gradient exchange.  Simple plans (fixed-length fields over distinct tables, the Criteo shape): one field per group, one
    id column per slot -- the layout of rounds 3-5, unchanged.  (Round 6: general groups; reference for what a group is:
    inputs.py:158-180.)"""

    def __init__(self, plan, world, rank):
        if plan.emb_dim <= 0 or not plan.deep:
            raise NotImplementedError("table-sharded training needs sparse features that share one embedding_dim")
        if not plan.unit_path:
            raise NotImplementedError("a table fed by more than 128 columns is beyond the deterministic update kernel")
        self.world, self.rank = int(world), int(rank)
        self.F, self.D = len(plan.deep), int(plan.emb_dim)
        # ---- groups: deep fields by table; the wide fields of the same features join their group
        by_table, groups = {}, []
        for i, f in enumerate(plan.deep):
            g = by_table.get(id(f.param))
            if g is None:
                g = by_table[id(f.param)] = len(groups)
                groups.append({"deep": [], "wide": [], "names": []})
            groups[g]["deep"].append(i)
            groups[g]["names"].append(f.name)
        name_group = {}
        for g, grp in enumerate(groups):
            for n in grp["names"]:
                name_group[n] = g
        wide_tables = {}
        for k, w in enumerate(plan.wide):
            g = name_group.get(w.name)
            if g is None:
                raise NotImplementedError("every linear sparse feature must also be a DNN feature for the sharded path")
            other = wide_tables.setdefault(id(w.param), g)
            if other != g:
                raise NotImplementedError("a linear table shared across the DNN's table groups cannot be sharded by table")
            groups[g]["wide"].append(k)
        n_wide_names = len(set(w.name for w in plan.wide))
        if plan.wide and n_wide_names != len(set(f.name for f in plan.deep)):
            raise NotImplementedError("linear sparse features must cover all or none of the DNN sparse features")
        self.has_wide = bool(plan.wide)
        self.groups = groups
        self.simple = bool(plan.simple_units)
        self.group_owner = self.assign(plan, groups, self.world)
        self.owner = [0] * self.F                      # per deep field
        for g, grp in enumerate(groups):
            for i in grp["deep"]:
                self.owner[i] = self.group_owner[g]
        # slot of a field = its position in its owner's SUB-PLAN (_hip/plan.py _fields_for: fixed-length fields first, then
        # the pooled ones, each in the model's order)
        self.slot = [0] * self.F
        self.fields_of = []                            # per rank: its deep fields in slot order
        for q in range(self.world):
            mine = [i for i in range(self.F) if self.owner[i] == q]
            mine = [i for i in mine if plan.deep[i].len == 1 and plan.deep[i].pool == 0] + \
                   [i for i in mine if not (plan.deep[i].len == 1 and plan.deep[i].pool == 0)]
            for s_, i in enumerate(mine):
                self.slot[i] = s_
            self.fields_of.append(mine)
        self.owned = list(self.fields_of[self.rank])
        self.n_slots = max(1, max(len(m) for m in self.fields_of))
        self.owner_slot = [self.owner[i] | (self.slot[i] << 16) for i in range(self.F)]   # what the assemble kernels read
        # X columns every destination needs from a sender: every position of the owner's fields (deep or wide) and their
        # length columns, each once, in slot order; [N * n_ids] (unused entries repeat column 0)
        self.cols_of, self.local_index = [], []
        for q in range(self.world):
            cols, fi_local = [], {}
            fields = [plan.deep[i] for i in self.fields_of[q]]
            names = set(f.name for f in fields)
            fields += [w for w in plan.wide if w.name in names]
            for f in fields:
                if f.name not in fi_local:
                    fi_local[f.name] = (len(cols), len(cols) + f.len)
                    cols += list(range(f.col, f.col + f.len))
                if f.len_col >= 0:
                    key = ("len", f.len_col)
                    if key not in fi_local:
                        fi_local[key] = (len(cols), len(cols) + 1)
                        cols.append(f.len_col)
            self.cols_of.append(cols)
            self.local_index.append(fi_local)
        self.n_ids = max(1, max(len(c) for c in self.cols_of))
        self.wide_col = (self.n_slots * self.D + 3) // 4 * 4
        self.ids_col = self.wide_col + 4
        self.ldc = (self.ids_col + self.n_ids + 3) // 4 * 4
        first = plan.deep[0].col
        cols = []
        for q in range(self.world):
            cols += self.cols_of[q] + [first] * (self.n_ids - len(self.cols_of[q]))
        self.id_cols = cols

    @staticmethod
    def assign(plan, groups, world):
        """owner of every table group: balanced twice over.  (1) SLOTS: a group brings one slot per deep field to its
        owner's chunk and B entries per X column to its gather and update; groups go, largest first, to the rank with the
        fewest slots so far -- for one-field groups (the Criteo shape) every rank then owns floor(F / N) or ceil(F / N)
        fields.  (2) BYTES: ties go to the rank that holds the fewest table bytes (LPT), so a 10 M-row table does not land
        beside another one while a rank of ten-row tables idles its HBM.  Deterministic: every rank computes the same map."""
        def nbytes(grp):
            p = plan.deep[grp["deep"][0]].param
            n = int(p.shape[0]) * int(p.shape[1])
            seen = set()
            for k in grp["wide"]:
                w = plan.wide[k].param
                if id(w) not in seen:
                    seen.add(id(w))
                    n += int(w.shape[0])
            return n

        def slots(grp):
            return len(grp["deep"])
        order = sorted(range(len(groups)), key=lambda g: (-slots(groups[g]), -nbytes(groups[g]), g))
        load, count, owner = [0] * world, [0] * world, [0] * len(groups)
        for g in order:
            q = min(range(world), key=lambda r: (count[r], load[r], r))
            owner[g] = q
            load[q] += nbytes(groups[g])
            count[q] += slots(groups[g])
        return owner

    def pack_ids(self, X, idx):
        """[N][B][n_ids] float ids: what each owner needs of this rank's B samples (``idx`` = id_cols on X's device)."""
        B = X.shape[0]
        return X.index_select(1, idx).view(B, self.world, self.n_ids).permute(1, 0, 2)


class HipShardOps(object):
    """The four device steps of the sharded exchange on the C-ABI (embed.hip / update.hip / shard.hip)."""

    def __init__(self, model, layout):
        from ._hip import lib as L
        from ._hip.plan import EmbeddingPlan
        from .inputs import SparseFeat, VarLenSparseFeat
        self.L, self.lay = L, layout
        plan = model.model_plan()
        self.plan = plan
        # the owner's SUB-PLAN: its deep fields in slot order (fixed-length first, then pooled -- the order EmbeddingPlan
        # itself gives them), the wide columns of the same features, over a compact id matrix [N * B, n_ids] whose columns
        # are layout.cols_of[rank]
        by_name = {}
        for c in model.dnn_feature_columns:
            if isinstance(c, (SparseFeat, VarLenSparseFeat)):
                by_name.setdefault(c.name, c)
        lin_by_name = {c.name: c for c in model._linear_feature_columns if isinstance(c, (SparseFeat, VarLenSparseFeat))}
        mine = [by_name[plan.deep[i].name] for i in layout.owned]
        fi_local = layout.local_index[layout.rank]
        fi = {}
        for c in mine:
            fi[c.name] = fi_local[c.name]
            ln = getattr(c, "length_name", None)
            if ln is not None:
                fi[ln] = fi_local[("len", model.feature_index[ln][0])]
        wide_cols = [lin_by_name[c.name] for c in mine if c.name in lin_by_name] if layout.has_wide else []
        self.sub = EmbeddingPlan(fi, deep_columns=mine, deep_tables=model.embedding_dict, wide_columns=wide_cols,
                                 wide_tables=model.linear_model.embedding_dict if wide_cols else None,
                                 wide_dense_weight=None, with_dense=False) if mine else None
        if self.sub is not None:
            self.sub.share_update_with(plan)
            got = [f.name for f in self.sub.deep]
            want = [plan.deep[i].name for i in layout.owned]
            if got != want:
                raise RuntimeError("the owner's sub-plan orders its fields %r, the layout expects %r" % (got, want))
        self._idx = None
        self._map = None
        self._idc = None
        self._side_bufs = {}        # general sub-plans: (den_t, amax) per global batch size, shared by gather and update

    def _owner_map(self, dev):
        if self._map is None or self._map.device != dev:
            self._map = torch.tensor(self.lay.owner_slot, dtype=torch.int32, device=dev)
        return self._map

    def _ptr(self, t, off=0):
        import ctypes
        return None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * off)

    def pack_ids(self, X):
        """[N][B][n_slots] float ids (a permuted view): what each owner needs of this rank's B samples."""
        lay = self.lay
        if self._idx is None or self._idx.device != X.device:
            self._idx = torch.tensor(lay.id_cols, dtype=torch.long, device=X.device)
        return lay.pack_ids(X, self._idx)

    def gather(self, ids_all, out=None, push=None):
        """ids_all [N*B, n_slots] (any row stride) -> (chunks [N*B, ldc] = rows of MY tables for every global sample,
        ids_t).  ``out`` = (chunks, ids_t, parts_t) preallocated (the whole-step graph of the direct exchange: what one
        replay gathers at its end, the next replay sends at its start).  ``push`` = (table, B): rank r's B rows are
        written straight into ``(float*)table[r]`` -- r's receive buffer -- instead of ``chunks`` (dctr_plan_t.out_chunks)."""
        L, lay, sub = self.L, self.lay, self.sub
        NB, dev = ids_all.shape[0], ids_all.device
        chunks = out[0] if out is not None else torch.empty((NB, lay.ldc), dtype=torch.float32, device=dev)
        if sub is None:
            return chunks, None
        cplan = sub.bind(dev)
        ids_t = out[1] if out is not None else torch.empty((sub.n_vcols, NB), dtype=torch.int32, device=dev)
        parts_t = out[2] if out is not None else torch.empty((sub.n_vcols, NB), dtype=torch.int16, device=dev)
        wide = self._ptr(chunks, lay.wide_col) if lay.has_wide else None
        sub.cplan.out_chunks = push[0].data_ptr() if push is not None else None
        sub.cplan.chunk_rows = int(push[1]) if push is not None else 0
        general = sub.gen is not None
        if general:
            # pooled / shared-table fields: the owner pools its positions locally (one row of D floats per field leaves);
            # mean pooling's divisors (written with the ids) and max pooling's arg-max positions (written by the gather) stay
            # here for the update of the same batch
            sub.point_step_buffers(*self._general_bufs(NB, dev))
        L.check(L.lib().dctr_embed_fwd(cplan, self._ptr(ids_all), ids_all.stride(0), NB, self._ptr(chunks), lay.ldc,
                                       wide, lay.ldc, None, self._ptr(self.plan.err_flag(dev)), sub.units_ptr(),
                                       sub.n_grid_units, None if general else self._ptr(ids_t),
                                       None if general else self._ptr(parts_t), None, 0, L.stream_handle(dev)),
                "dctr_embed_fwd(owned tables, global batch)")
        if general:
            L.check(L.lib().dctr_embed_ids(cplan, sub.units_ptr(), sub.n_grid_units, self._ptr(ids_all), ids_all.stride(0), NB,
                                           self._ptr(ids_t), self._ptr(parts_t), L.stream_handle(dev)), "dctr_embed_ids")
        # the id-only half of the owners' update (find + sort every partition's entries) starts now, on a side stream,
        # under the row all-to-all and the tower (EmbeddingPlan.launch_segments)
        # (opt-in: measured at one rank the step is host-paced -- the side-stream hand-offs of the pre-pass cost more
        # host time than the update kernel saves: 0.290 vs 0.233 ms per step)
        handle = sub.launch_segments(ids_t, parts_t, NB) if (os.environ.get("DCTR_SHARDED_SEGMENTS", "0") == "1" and
                                                             dev.type == "cuda") else None
        return chunks, (ids_t, parts_t, handle)

    def _general_bufs(self, NB, dev):
        key = (int(NB), str(dev))
        b = self._side_bufs.get(key)
        if b is None:
            b = self._side_bufs[key] = self.sub.step_buffers(NB, dev)
        return b

    def assemble_fwd(self, recv, X, want_fm):
        L, lay, plan = self.L, self.lay, self.plan
        B, dev = X.shape[0], X.device
        plan.bind(dev)
        out = torch.empty((B, plan.ld_out), dtype=torch.float32, device=dev)
        wide = torch.empty((B,), dtype=torch.float32, device=dev) if plan.has_wide else None
        fm = torch.empty((B,), dtype=torch.float32, device=dev) if want_fm else None
        ld_s = (lay.D + 3) // 4 * 4
        fm_s = torch.empty((B, ld_s), dtype=torch.float32, device=dev) if want_fm else None
        w = plan.wide_dense_weight
        L.check(L.lib().dctr_shard_assemble_fwd(
            self._ptr(recv), lay.ldc, lay.world, B, lay.F, lay.D, self._ptr(self._owner_map(dev)),
            lay.wide_col if lay.has_wide else -1, self._ptr(X), X.stride(0), self._ptr(plan._dev["dense"]), len(plan.dense_cols), max(plan.dense_off, 0),
            self._ptr(plan._dev["wdense"]), self._ptr(w), len(plan.wdense_cols) if w is not None else 0,
            self._ptr(out), plan.ld_out, self._ptr(wide), self._ptr(fm), self._ptr(fm_s), ld_s,
            L.stream_handle(dev)), "dctr_shard_assemble_fwd")
        return out, wide, fm, fm_s

    def assemble_bwd(self, X, g_out, g_wide, g_fm, out, fm_s, g_wdense, send=None, push=None, next_x=None, carry=True):
        """``push``: device table of the owners' receive buffers -- chunk q is written there, not into ``send``.
        ``next_x`` (push only): the NEXT batch's input matrix -- its id columns ride along straight from there
        (dctr_shard_assemble_bwd_next; otherwise from the local send rows dctr_shard_stage filled); ``carry=False``: nothing
        rides along (the next batch is not announced)."""
        L, lay, plan = self.L, self.lay, self.plan
        B, dev = X.shape[0], X.device
        if send is None:
            send = torch.empty((lay.world * B, lay.ldc), dtype=torch.float32, device=dev)
        w = plan.wide_dense_weight
        carry_n = lay.n_ids if (push is not None and carry) else 0
        if next_x is not None and (self._idc is None or self._idc.device != dev):
            self._idc = torch.tensor(lay.id_cols, dtype=torch.int32, device=dev)
        L.check(L.lib().dctr_shard_assemble_bwd_next(
            self._ptr(send), self._ptr(push), lay.ids_col if push is not None else 0, carry_n,
            lay.ldc, lay.world, B, lay.F, lay.D, self._ptr(self._owner_map(dev)),
            lay.wide_col if lay.has_wide else -1, self._ptr(g_out), g_out.stride(0) if g_out is not None else 0, self._ptr(g_wide), self._ptr(g_fm),
            self._ptr(out), plan.ld_out, self._ptr(fm_s), fm_s.stride(0) if fm_s is not None else 0, self._ptr(X),
            X.stride(0), self._ptr(plan._dev["wdense"]), len(plan.wdense_cols) if (w is not None and g_wdense is not None) else 0,
            self._ptr(g_wdense), self._ptr(next_x) if (next_x is not None and carry_n) else None,
            next_x.stride(0) if (next_x is not None and carry_n) else 0,
            self._ptr(self._idc) if (next_x is not None and carry_n) else None, L.stream_handle(dev)), "dctr_shard_assemble_bwd")
        return send

    def update(self, grads_all, ids_t):
        """grads_all [N*B, ldc]: apply the fused optimizer step to MY tables over the global batch."""
        L, lay, sub, plan = self.L, self.lay, self.sub, self.plan
        if sub is None:
            return
        NB, dev = grads_all.shape[0], grads_all.device
        kind = plan.update[0]
        if kind in ("sgd", "sgd2"):
            opt, lr, eps = L.UPD_SGD, float(plan.update[1]), 0.0
        elif kind == "adagrad":
            opt, lr, eps = L.UPD_ADAGRAD, float(plan.update[1]), float(plan.update[2])
        else:
            raise RuntimeError("table-sharded training needs the fused sparse update (compile('sgd' | 'adagrad'), "
                               "l2_reg_embedding = l2_reg_linear = 0)")
        cplan = sub.bind(dev)
        if not sub.update_kernel_ok(NB):
            raise RuntimeError("global batch %d is beyond the deterministic update kernel" % NB)
        gw = self._ptr(grads_all, lay.wide_col) if lay.has_wide else None
        ids_t, parts_t, handle = ids_t
        ws, ws_n, pre = sub.update_workspace_for(ids_t, handle, NB)
        if sub.gen is not None:
            sub.point_step_buffers(*self._general_bufs(NB, dev))     # (what this batch's gather left: divisors, arg-max)
        L.check(L.lib().dctr_embed_update(cplan, sub.units_ptr(), sub.n_grid_units, sub.max_vocab, self._ptr(ids_t),
                                          self._ptr(parts_t), NB,
                                          self._ptr(grads_all), lay.ldc, None, 0, None, 0, None, gw, lay.ldc, opt, lr,
                                          eps, None, 0, None, None, self._ptr(ws), ws_n, pre, L.stream_handle(dev)),
                "dctr_embed_update(owned tables)")


class DirectExchange(object):
    """The sharded step's three exchanges without a host-issued collective: every rank owns receive buffers that the OTHER
    ranks' copies write into -- device memory mapped into every process through torch's CUDA IPC -- and one arrival word per
    (exchange kind, sender) (include/dctr.h, dctr_exchange_post / _wait / _next).

      rows    owner q's chunk for rank r  -> r's ``recv[q]``    (the forward all-to-all)
      grads   rank r's chunk for owner q  -> q's ``grads[r]``   (the sparse reduce-scatter; carries the next batch's ids)
      dense   rank r's dense gradient slab -> everybody's ``dense[r]``, then a local sum in rank order (all-reduce as
              all-gather + sum: N x the bytes of a ring, 0.57 MB x 7 on links that move 50+ GB/s each, and bit-identical
              sums on every rank by construction)

    One-shot and point-to-point: N - 1 copies per exchange, one per peer, all links busy at once (xGMI is a full mesh of
    point-to-point links).  Everything is an ordinary stream operation -- copies, a one-wave post kernel, a
    one-wave wait kernel -- so the WHOLE train step including its exchanges is captured into one hipGraph: the host issues
    one graph launch per step where the RCCL path issues three collectives and four segments (~0.22 ms of host time per
    step at one rank, profiles/r04_bench_sharded_1rank_rccl.json: that path is host-paced).

    Buffer reuse needs no extra handshake: owner q can only produce rows for step k + 1 after its update of step k, which
    needs every rank's gradients of step k, which every rank sends after it has read its rows of step k; words only grow."""
    KINDS = {"rows": 0, "grads": 1, "dense": 2, "ids": 3}

    def __init__(self, group, world, rank, device, B, ldc, n_slots, n_dense, timeout_us=5000000, dense_src=None, push=None):
        """``push`` (default: env DCTR_SHARDED_PUSH != "0"): the producing KERNELS write into the peers' buffers (tables
        ``row_tbl`` / ``grad_tbl`` of this rank's slots there) and the dense sum reads the peers' gradient slabs
        (``dense_src``, shared like the receive buffers) where they lie -- no copy nodes at all.  Otherwise every
        exchange is N copies out of a local staging buffer."""
        from torch.multiprocessing.reductions import reduce_tensor
        from ._hip import lib as L
        self.L, self.group, self.world, self.rank = L, group, int(world), int(rank)
        self.timeout_us = int(timeout_us)
        dev = torch.device(device)
        f32 = dict(dtype=torch.float32, device=dev)
        self.ld_dense = (int(n_dense) + 3) // 4 * 4
        self.recv = torch.zeros((world, B, ldc), **f32)
        self.grads = torch.zeros((world, B, ldc), **f32)
        self.dense = torch.zeros((world, self.ld_dense), **f32)
        self.ids = torch.zeros((world, B, n_slots), **f32)
        self.words = torch.zeros((len(self.KINDS), 64), dtype=torch.int32, device=dev)
        # one exchange counter per kind, advanced by that kind's own sync: exchanges of different kinds may sit on
        # different queues (the dense exchange runs beside the gradient exchange and the owners' update)
        self.step = torch.ones((len(self.KINDS),), dtype=torch.int32, device=dev)
        self.err = torch.zeros((1,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        self.push = (os.environ.get("DCTR_SHARDED_PUSH", "1") != "0") if push is None else bool(push)
        mine = [self.recv, self.grads, self.dense, self.ids, self.words]
        if self.push and dense_src is not None:
            mine.append(dense_src)
        self.peers = [[None] * world for _ in mine]
        if world == 1:
            for k, t in enumerate(mine):
                self.peers[k][0] = t
        else:
            handles = [reduce_tensor(t) for t in mine]
            got = [None] * world
            dist.all_gather_object(got, [h[1] for h in handles], group=group)
            for r in range(world):
                for k, t in enumerate(mine):
                    self.peers[k][r] = t if r == rank else handles[k][0](*got[r][k])
            self._keep = got
            # this device's kernels and copies reach every peer's buffers (ranks on ONE device: nothing to enable)
            with torch.cuda.device(dev):
                for r in range(world):
                    pd = self.peers[0][r].device.index
                    if pd is not None and pd != dev.index:
                        L.check(L.lib().dctr_enable_peer_access(int(pd)), "dctr_enable_peer_access(%d)" % pd)
        # device tables of the peers' word arrays, one per exchange kind
        self.word_ptrs = []
        for kind in range(len(self.KINDS)):
            self.word_ptrs.append(torch.tensor([self.peers[4][r].data_ptr() + 4 * 64 * kind for r in range(world)],
                                               dtype=torch.int64, device=dev))
        # push-style: this rank's slot in every peer's receive buffers; pull: every peer's dense gradient slab
        i64 = dict(dtype=torch.int64, device=dev)
        self.row_tbl = torch.tensor([self.peers[0][r][rank].data_ptr() for r in range(world)], **i64)
        self.grad_tbl = torch.tensor([self.peers[1][r][rank].data_ptr() for r in range(world)], **i64)
        self.dense_tbl = torch.tensor([self.peers[5][r].data_ptr() for r in range(world)], **i64) if len(mine) > 5 else None
        if world > 1:
            dist.barrier(group=group)

    def _ptr(self, t, off=0):
        import ctypes
        return ctypes.c_void_p(t.data_ptr() + off)

    def _sync(self, kind):
        """post this rank's arrival to every peer, wait for every peer's; the kind's counter moves on behind it"""
        L, k = self.L, self.KINDS[kind]
        L.check(L.lib().dctr_exchange_sync(self._ptr(self.word_ptrs[k]), self._ptr(self.words, 4 * 64 * k), self.world,
                                           self.rank, self._ptr(self.step, 4 * k), 1, self.timeout_us, self._ptr(self.err),
                                           L.stream_handle(self.words.device)),
                "dctr_exchange_sync")

    def _copy(self, dst, src):
        """device-to-device copy on the current stream as ONE hipMemcpyAsync (capturable; no torch cross-device stream
        logic: the destination may live on a peer GPU)"""
        if not (dst.is_contiguous() and src.is_contiguous()) or dst.numel() != src.numel() or dst.dtype != src.dtype:
            dst.copy_(src, non_blocking=True)
            return
        L = self.L
        L.check(L.lib().dctr_copy_async(self._ptr(dst), self._ptr(src), dst.numel() * dst.element_size(),
                                        L.stream_handle(src.device)), "dctr_copy_async")

    def _scatter(self, which, src):
        """src [world, ...]: slice r goes to rank r's buffer `which`, into this rank's slot"""
        for r in range(self.world):
            self._copy(self.peers[which][r][self.rank], src[r])

    def send_rows(self, chunks):
        """(push: the gather has written the peers' buffers already -- HipShardOps.gather(push=...))"""
        if not self.push:
            self._scatter(0, chunks.view(self.recv.shape))
        self._sync("rows")
        return self.recv.view(-1, self.recv.shape[2])

    def send_grads(self, send):
        if not self.push:
            self._scatter(1, send.view(self.grads.shape))
        self._sync("grads")
        return self.grads.view(-1, self.grads.shape[2])

    def send_ids(self, ids):
        self._scatter(3, ids)
        self._sync("ids")
        return self.ids.view(-1, self.ids.shape[2])

    def allreduce_dense(self, flat, step=None):
        """flat <- sum over ranks (rank order); ``step`` (dctr_dense_step_t): the sum kernel also applies the optimizer step."""
        import ctypes
        n = flat.numel()
        L = self.L
        if self.dense_tbl is not None and step is not None:
            # pull: the peers' slabs are read where they lie.  Safe against the peers' next step: rank p overwrites its
            # slab only behind its next rows wait, which needs every rank's next rows post -- issued behind that rank's sum.
            self._sync("dense")
            L.check(L.lib().dctr_sum_ranks(self._ptr(flat), None, self._ptr(self.dense_tbl), self.world, n, 0,
                                           ctypes.byref(step), 0, L.stream_handle(flat.device)), "dctr_sum_ranks(pull)")
            return
        for r in range(self.world):
            self._copy(self.peers[2][r][self.rank][:n], flat)
        self._sync("dense")
        L.check(L.lib().dctr_sum_ranks(self._ptr(flat), self._ptr(self.dense), None, self.world, n, self.ld_dense,
                                       ctypes.byref(step) if step is not None else None, 1, L.stream_handle(flat.device)),
                "dctr_sum_ranks")

    def self_test(self, rounds=3):
        """Does the direct exchange move the right BYTES between these ranks?  Model-free, a few hundred microseconds:
        ``rounds`` times (the same addresses, different values -- a stale line in anybody's cache would show) every rank
        (1) stores a pattern that names (round, sender, receiver, element) into its slot of every peer's ``recv`` buffer with
        a KERNEL running on its own device (dctr_sum_ranks over one source: the way the owners' gather and the gradient
        assembly push), (2) passes the rows exchange, (3) compares what arrived from every sender with what that sender must
        have written, (4) pulls every peer's ``dense`` slab through the pointer table (the way the dense sum reads them) and
        compares the sum, (5) passes the gradient exchange so that nobody overwrites a buffer somebody still checks.
        Returns the number of wrong elements seen by THIS rank (+ 1 when a wait timed out); the caller reduces over ranks.
        bench.py / distributed_fit use it before they trust the exchange on more than one GPU: it has only ever been
        exercised between processes that share one device (tests/test_gpu_direct_exchange.py)."""
        import ctypes
        L = self.L
        dev = self.recv.device
        n = int(self.recv.shape[1] * self.recv.shape[2])
        nd = int(self.ld_dense)
        idx = torch.arange(n, dtype=torch.float32, device=dev) * (1.0 / 1024.0)
        idd = torch.arange(nd, dtype=torch.float32, device=dev) * (1.0 / 1024.0)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        stage = torch.empty((self.world, n), dtype=torch.float32, device=dev)
        pulled = torch.empty((nd,), dtype=torch.float32, device=dev)
        tbl = torch.tensor([self.peers[2][r][r].data_ptr() for r in range(self.world)], dtype=torch.int64, device=dev)
        stream = L.stream_handle(dev)
        for it in range(int(rounds)):
            for q in range(self.world):
                stage[q] = idx + float(1000 * it + 10 * self.rank + q)
            self.dense[self.rank, :nd] = idd + float(100 * it + self.rank)        # my own slab, read by everybody in (4)
            for q in range(self.world):
                dst = self.peers[0][q][self.rank]
                L.check(L.lib().dctr_sum_ranks(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(stage[q].data_ptr()), None,
                                               1, n, n, None, 1, stream), "dctr_sum_ranks(push)")
            self._sync("rows")
            for r in range(self.world):
                want = idx + float(1000 * it + 10 * r + self.rank)
                bad += (self.recv[r].reshape(-1) != want).sum()
            L.check(L.lib().dctr_sum_ranks(ctypes.c_void_p(pulled.data_ptr()), None, ctypes.c_void_p(tbl.data_ptr()),
                                           self.world, nd, 0, None, 1, stream), "dctr_sum_ranks(pull)")
            want = torch.zeros_like(pulled)
            for r in range(self.world):           # (rank order, like the kernel: the same float sums)
                want = want + (idd + float(100 * it + r))
            bad += (pulled != want).sum()
            self._sync("grads")
        torch.cuda.synchronize(dev)
        n_bad = int(bad.item())
        if int(self.err.item()) != 0:
            self.err.zero_()
            n_bad += 1
        return n_bad

    def check(self):
        """Raise if a wait ever timed out (synchronises the device)."""
        if int(self.err.item()) != 0:
            self.err.zero_()
            raise RuntimeError("a direct-exchange wait timed out: a peer rank did not reach the same exchange "
                               "(ranks out of step, or a rank died)")


def resolve_exchange(requested, device, group=None, verbose=True):
    """'rccl' | 'direct' | 'auto' | 'try-direct'  ->  ('rccl' | 'direct', what happened).

    'auto' (the default of ``fit()`` under torchrun and of ``bench.py``): the direct exchange wherever it can be TRUSTED -- one
    rank always; several ranks after ``DirectExchange.self_test`` has passed on every one of them (a model-free check of the
    bytes that kernels push into / pull out of the peers' IPC-mapped buffers, the arrival words and the time-outs); RCCL
    otherwise (CPU stand-ins, a failed or throwing self-test on ANY rank).  Collective: every rank must call it."""
    dev = torch.device(device)
    world = dist.get_world_size(group)
    if requested in ("rccl", "direct"):
        return requested, "as requested"
    if requested not in ("auto", "try-direct"):
        raise ValueError("exchange must be 'rccl', 'direct', 'auto' or 'try-direct'")
    if dev.type != "cuda":
        return "rccl", "no GPU: torch.distributed collectives"
    if world == 1:
        return "direct", "one rank"
    ok, why = 1, ""
    try:
        dx = DirectExchange(group, world, dist.get_rank(group), dev, 64, 32, 4, 1024,
                            dense_src=torch.zeros(1024, dtype=torch.float32, device=dev), timeout_us=2000000)
        n_bad = dx.self_test()
        if n_bad:
            ok, why = 0, "%d wrong elements / timed-out waits" % n_bad
        del dx
    except Exception as exc:          # (IPC mapping refused, peer access missing, a launch error ...)
        ok, why = 0, "%s: %s" % (type(exc).__name__, str(exc)[:200])
    flag = torch.tensor([ok], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    passed = int(flag.item()) == 1
    if not ok and verbose:
        import sys
        print("direct exchange self-test failed on rank %d (%s)" % (dist.get_rank(group), why), file=sys.stderr, flush=True)
    return ("direct", "self-test passed on every rank") if passed else ("rccl", "self-test failed on at least one rank")


class _GradTap(torch.autograd.Function):
    """identity whose backward stores the incoming gradient in ``box[key]`` and ends the graph there"""

    @staticmethod
    def forward(ctx, x, box, key):
        ctx.box, ctx.key = box, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.box[ctx.key] = g
        return None, None, None


class _Segment(object):
    """A piece of the step between two collectives: run eagerly, or captured once into a hipGraph and replayed
    (collectives stay OUTSIDE the graphs: the host issues 4 graph launches + 3 collectives per step instead of
    ~40 kernel launches through Python)."""

    def __init__(self, fn, use_graph):
        self.fn, self.use_graph = fn, use_graph
        self.graph, self.result = None, None
        self.primed = False

    def __call__(self):
        if not self.use_graph:
            return self.fn()
        if not self.primed:         # first call runs eagerly: descriptor uploads / lazy buffers are not capturable
            self.primed = True
            return self.fn()
        if self.graph is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: RCCL's watchdog thread keeps polling events while we capture
            from ._hip.graph import no_gc_during_capture
            from ._hip import streams as _streams
            dev = torch.device("cuda", torch.cuda.current_device())
            with no_gc_during_capture(), torch.cuda.graph(g, capture_error_mode="thread_local",
                                                          stream=_streams.side_stream(dev, "capture")):
                self.result = self.fn()
            self.graph = g
        self.graph.replay()
        return self.result


class ShardedTrainer(object):
    """Multi-GPU training with table-sharded embeddings, one process per GPU: models on the fused train step
    (``BaseModel._fused_step_state``: DeepFM, WDL, NFM) and -- since round 3 -- every other model whose tables take the
    fused sparse update (xDeepFM, FiBiNET, DCN, PNN, ...: the model's own forward / autograd / optimizer between the same
    exchange steps).

      tables   sharded by table GROUP (a deep table with every feature column that feeds it -- fixed-length, pooled VarLen,
               shared through embedding_name -- and the same features' wide tables; ``layout.group_owner[g]`` owns group g and
               its Adagrad state; balanced by slots, then bytes: ShardLayout.assign);
               forward = owner-side gather + rows all-to-all, backward = row-gradient all-to-all (the sparse
               reduce-scatter, which also carries the NEXT batch's ids to the owners) + the owner's deterministic
               fused update.  Each row is updated once, by its owner, from the gradients of ALL N*B samples -- the
               same step as one GPU on the global batch.
      dense    replicated; ONE all-reduce(SUM) of the flat gradient slab (SUM: the loss is a sum over the global
               batch, basemodel.py:209,254), overlapped with the sparse update; every rank then applies the same
               fused dense optimizer step, so replicas stay bit-identical.

    Per step: ``[gather] -a2a-> [assemble, tower, head, tower backward, assemble^T] -a2a-> [update] || all-reduce ->
    [dense step]``: four compute segments and three collectives (an extra ids all-to-all only when the caller did
    not announce the batch in the previous step's ``next_xb``).  With ``use_graphs`` each segment is one hipGraph
    (fixed batch shape), the collectives are issued by the host between the replays.  (Capturing the RCCL
    collectives inside one whole-step graph was tried on MI355X / ROCm 7.2 / RCCL 2.26 and deadlocks at the first
    replay, so collectives stay outside the graphs.)

    Only the owner's copy of a table is current during training; ``gather_tables()`` broadcasts the owners' copies
    (and optimizer state) so that ``state_dict()`` is complete on every rank -- call it before saving / predicting."""

    def __init__(self, model, process_group=None, ops=None, broadcast_parameters=True, use_graphs=False, exchange=None):
        if not dist.is_initialized():
            raise RuntimeError("initialise torch.distributed first (backend 'nccl' = RCCL on ROCm)")
        self.model, self.group = model, process_group
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.plan = model.model_plan()
        if broadcast_parameters:
            with torch.no_grad():
                for p in model.parameters():
                    _broadcast(p.data, 0, process_group)
        if self.plan.update[0] not in ("sgd", "adagrad"):
            raise NotImplementedError(
                "table-sharded training needs the fused sparse table update: compile('sgd' | 'adagrad') with "
                "l2_reg_embedding = l2_reg_linear = 0 (the lazy regularised / Adam update is single-GPU; "
                "DataParallelTrainer takes any optimizer on replicated tables)")
        st = model._fused_step_state()
        if st is not None and st["slab"].lam is not None:
            st = None
        # Two routes for everything that is not a table:
        #   fused     (DeepFM, WDL, NFM: BaseModel._fused_step_state) tower + head + backward as kernels, dense gradients in
        #             the DenseSlab, one fused dense optimizer launch;
        #   autograd  (xDeepFM, FiBiNET, DCN, PNN, AFM, AutoInt, ...: round 3) the model's own forward over the assembled
        #             rows, autograd into ONE flat gradient bucket, the model's own optimizer over the dense parameters.
        # The exchange -- owners' gather, two all-to-alls, owners' fused update -- is the same for both.
        self.state = st
        self.slab = st["slab"] if st is not None else None
        self.bucket = None
        if st is None:
            tables = set(id(p) for p in self.plan.table_params)
            self.bucket = DenseBucket([p for p in model.parameters() if id(p) not in tables])
            self._bucket_view = {id(p): v for p, v in zip(self.bucket.params, self.bucket.views)}
        self.layout = ShardLayout(self.plan, self.world, self.rank)
        self.ops = ops if ops is not None else HipShardOps(model, self.layout)
        self.use_graphs = bool(use_graphs)
        # (opt-in: at one rank, where the all-to-alls are short, the extra stream hand-offs cost more than the overlap
        # gains -- 255 vs 234 us per step on MI355X; the multi-rank effect is unmeasured)
        self.overlap_wgrad = os.environ.get("DCTR_SHARDED_OVERLAP_WGRAD", "0") == "1"
        # "rccl": torch.distributed collectives issued by the host between hipGraph segments; "direct": copies into the
        # peers' IPC-mapped buffers + arrival words, the whole step one hipGraph (DirectExchange; fused route, GPU only)
        self.exchange = exchange or os.environ.get("DCTR_SHARDED_EXCHANGE", "rccl")
        self.exchange_note = "as requested"
        if self.exchange in ("auto", "try-direct"):
            # (the direct exchange drives the fused route only; the autograd route keeps the host-issued collectives)
            if st is None:
                self.exchange, self.exchange_note = "rccl", "autograd route"
            else:
                self.exchange, self.exchange_note = resolve_exchange(self.exchange, model.device, process_group)
        if self.exchange not in ("rccl", "direct"):
            raise ValueError("exchange must be 'rccl', 'direct', 'auto' or 'try-direct'")
        self._dx = None
        self._side = None
        self._pre = None          # (chunks, ids) of the announced next batch, gathered at the end of the previous call
        self._shape = None
        self._leaves = None
        self.plan.sharder = self

    def close(self):
        self._join()
        if self._dx is not None:
            self._dx.check()
        self.plan.sharder = None

    def set_use_graphs(self, on):
        """Switch hipGraph capture of the compute segment on / off; takes effect at the next ``train_step`` (the segments are
        rebuilt).  Eager steps first, graphs afterwards is the intended order: the first call of a segment uploads
        descriptors and allocates lazily, neither of which can be captured."""
        if bool(on) == bool(self.use_graphs):
            return        # (fit() asks at the start of every call: captured segments and blocks of an earlier call stay valid)
        self.use_graphs = bool(on)
        self._shape = None
        if self._dx is not None:
            self._direct_seg = _Segment(self._direct_body, bool(on))

    # ---- called by _ops.embed (from model.logit_parts) in place of the single-GPU lookup ----------------------------
    def embed(self, X, want_fm, full):
        plan = self.plan
        if not torch.is_grad_enabled():
            raise RuntimeError("a table-sharded model predicts after trainer.gather_tables(); trainer.close()")
        out, wide, fm, fm_s = self.ops.assemble_fwd(self._recv, X, want_fm)
        leaves = {"out": out, "fm_s": fm_s, "want_fm": bool(want_fm), "wide": wide, "fm": fm, "grads": {}}
        # the model sees taps of the three tensors: their backward hands the incoming gradient to this trainer as it is
        # (a leaf's .grad is a CLONE whenever somebody else still holds the gradient -- the tower keeps its gx alive for
        # the deferred weight gradients: a 7 MB copy node + a graph edge on the step's critical chain)
        out = _GradTap.apply(out.requires_grad_(), leaves["grads"], "out")
        wide = _GradTap.apply(wide.requires_grad_(), leaves["grads"], "wide") if wide is not None else None
        fm = _GradTap.apply(fm.requires_grad_(), leaves["grads"], "fm") if fm is not None else None
        self._leaves = leaves
        B = X.shape[0]
        o = out if (full or not plan.has_lookup) else out[:, :plan.width]
        return (o, wide if wide is not None else X.new_zeros((B,)), fm if fm is not None else X.new_zeros((B,)))

    # ---- the segments -----------------------------------------------------------------------------------------------
    def _build(self, xb, yb):
        lay, dev, B = self.layout, xb.device, xb.shape[0]
        self._shape = (tuple(xb.shape), tuple(yb.shape))
        self._x, self._y = torch.empty_like(xb), torch.empty_like(yb)
        eng = getattr(self, "_eng", None)
        if eng is not None:       # (the engine compute segment reads the staged batch out of its own [B, C + 1] buffer)
            if eng["xfull"].shape[0] == B and eng["xfull"].shape[1] == xb.shape[1] + 1 and eng["xfull"].device == xb.device:
                self._x = eng["xfull"][:, :xb.shape[1]]
            else:
                self._eng = None
        # captured direct-exchange segments hold the raw addresses of the staging tensors just replaced (same B, another
        # y shape: round-4 advisor finding) -- they are rebuilt with the new ones
        if getattr(self, "_dx", None) is not None:
            self._direct_seg = _Segment(self._direct_body, bool(self.use_graphs))
        self._blk = None
        self._ids_next = torch.zeros((lay.world, B, lay.n_ids), dtype=torch.float32, device=dev)
        self._ids_tmp = torch.empty((lay.world, B, lay.n_ids), dtype=torch.float32, device=dev)
        self._recv = torch.empty((lay.world * B, lay.ldc), dtype=torch.float32, device=dev)
        # what arrives in the gradient all-to-all: row gradients of step k AND the ids of step k+1
        self._grads_all = torch.zeros((lay.world * B, lay.ldc), dtype=torch.float32, device=dev)
        self._ids_view = self._grads_all[:, lay.ids_col:lay.ids_col + lay.n_ids]
        self._announced = None          # identity of the batch whose ids already sit in _ids_view
        self._ids_t = None
        g = bool(self.use_graphs) and xb.is_cuda and self.slab is not None    # (the autograd route runs eagerly)
        # Only the multi-kernel segment is a graph by default: a hipGraph launch leaves the GPU idle for ~12 us before its
        # first kernel, a plain launch ~2 us.  DCTR_SHARDED_GRAPH_ALL=1 captures the single-kernel segments too: their host
        # cost drops from 51 + 49 us to 9 + 7 us (round 3, profiles/r03_shard_host_profile_1rank.txt), and the step time
        # does not move (0.2305 vs 0.2315 ms): the host then waits that much longer inside the collectives -- the step
        # is paced by the GPU-side chain of kernels, collectives and stream hand-offs, not by the host.
        gs = g and os.environ.get("DCTR_SHARDED_GRAPH_ALL", "0") == "1"
        self._segB = _Segment(lambda: self.ops.gather(self._ids_view), gs)
        self._segC = _Segment(self._compute if self.slab is not None else self._compute_autograd, g)
        self._segD = _Segment(lambda: self.ops.update(self._grads_all, self._ids_t), gs)
        self._segE = _Segment((lambda: self.slab.step(*self.state["mode"])) if self.slab is not None
                              else self._dense_step_autograd, False)

    def _compute_autograd(self):
        """The middle segment for a model outside the fused step: its own forward on the assembled rows (``embed`` below
        is what ``logit_parts`` reaches through ``_ops.embed``), loss (sum) + the regularisers' share, autograd into the
        flat bucket; the gradients of the assembled rows / wide logit / FM input go back through ``assemble_bwd``."""
        model, plan, lay = self.model, self.plan, self.layout
        self._leaves = None
        y_pred = model(self._x).squeeze()
        model.optim.zero_grad()
        self.bucket.attach()
        if isinstance(model.loss_func, list):
            loss = sum([model.loss_func[i](y_pred[:, i], self._y[:, i], reduction='sum') for i in range(model.num_tasks)])
        else:
            loss = model.loss_func(y_pred, self._y.squeeze(), reduction='sum')
        # the dense parameters are replicated and their gradients SUM-all-reduced: terms that depend on parameters only
        # enter 1 / world per rank (tables carry no regulariser here: plan.update is the fused sparse update)
        reg = model.get_regularization_loss() + model.aux_loss
        (loss + reg * (1.0 / self.world)).backward()
        lv = self._leaves
        if lv is None:
            raise RuntimeError("the model's forward did not go through the fused lookup")
        w = plan.wide_dense_weight
        g_wd = self._bucket_view.get(id(w)) if w is not None else None
        g_out, g_wide, g_fm = self._tapped(lv)
        send = self.ops.assemble_bwd(self._x, g_out, g_wide, g_fm, lv["out"].detach(), lv["fm_s"],
                                     g_wd.reshape(-1) if (g_wd is not None and g_wide is not None) else None)
        B = self._x.shape[0]
        send.view(lay.world, B, lay.ldc)[:, :, lay.ids_col:lay.ids_col + lay.n_ids].copy_(self._ids_next)
        # (what the trainer logs as the total: loss + the FULL regularisation / auxiliary terms, like
        # DataParallelTrainer and the single-GPU step -- not this rank's 1 / world share that entered the backward)
        self._autograd_total = (loss + reg).detach().reshape(1)
        return send, loss.detach(), y_pred.detach()

    def _dense_step_autograd(self):
        model = self.model
        model._step_stacked_groups()
        done, _ = model._step_dense_multi(None)
        if not done:
            model.optim.step()

    @staticmethod
    def _tapped(lv):
        """(g_out, g_wide, g_fm) as the backward delivered them to the taps of ``embed`` (None: no gradient arrived)"""
        def pick(key, like):
            g = lv["grads"].get(key)
            if g is None or like is None:
                return None
            if g.dtype != torch.float32 or (g.dim() == 2 and (g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16)) or \
                    (g.dim() == 1 and g.numel() > 1 and g.stride(0) != 1):
                g = g.float().contiguous()
            return g
        return pick("out", lv["out"]), pick("wide", lv["wide"]), pick("fm", lv["fm"])

    def _compute(self):
        model, st, slab, plan, lay = self.model, self.state, self.slab, self.plan, self.layout
        model._grad_sink = slab
        self._leaves = None
        # the tower's weight gradients are not enqueued here: TowerHeadFunction leaves a closure in slab.deferred and
        # train_step() runs it on a second stream behind this segment, beside the gradient all-to-all and the update
        slab.overlap = "defer" if (self._x.device.type == "cuda" and self.overlap_wgrad) else False
        slab.deferred = None
        try:
            loss, y_pred = model.fused_loss(self._x, self._y, slab)
            loss.backward(gradient=st["one"])
        finally:
            model._grad_sink = None
            slab.overlap = False
        lv = self._leaves
        if lv is None:
            raise RuntimeError("the model's logit_parts() did not go through the fused lookup")
        g_wd = slab.grad_of(plan.wide_dense_weight) if plan.wide_dense_weight is not None else None
        g_out, g_wide, g_fm = self._tapped(lv)
        static = getattr(self, "_send_static", None)
        if static is not None:
            # (the direct-exchange step: the staging launch in front of the step has put the next batch's ids in place)
            send = self.ops.assemble_bwd(self._x, g_out, g_wide, g_fm, lv["out"].detach(), lv["fm_s"],
                                         g_wd if g_wide is not None else None, send=static,
                                         push=self._dx.grad_tbl if self._dx.push else None)
            return send, loss.detach(), y_pred
        send = self.ops.assemble_bwd(self._x, g_out, g_wide, g_fm, lv["out"].detach(), lv["fm_s"],
                                     g_wd if g_wide is not None else None)
        B = self._x.shape[0]
        send.view(lay.world, B, lay.ldc)[:, :, lay.ids_col:lay.ids_col + lay.n_ids].copy_(self._ids_next)
        return send, loss.detach(), y_pred

    def train_step(self, xb, yb, next_xb=None):
        """One optimizer step on the gradient of the loss summed over every rank's batch.  ``next_xb`` (optional):
        the batch of the NEXT call -- its ids are shipped to the owners together with this step's row gradients,
        which saves that step's ids all-to-all.  Every rank must announce (or not) consistently.

        Queues on a GPU (round 4; profiles/r03_sharded_1rank_timeline.txt showed one queue, every kernel serial, and
        ~40 % of the step idle between host-issued pieces):
          main   rows all-to-all -> [assemble, tower + head + backward-data, assemble^T] -> gradient all-to-all -> owners'
                 update -> owners' gather for the NEXT batch (its ids arrived with the gradients)
          side   the tower's weight gradients + reduction -> all-reduce of the dense gradient slab -> dense optimizer step
                 (beside the gradient all-to-all and the update; joined in front of the next tower launch)
          copy   this call's batch and the announced next batch's ids into the static buffers the captured segment reads
                 (beside the rows all-to-all)"""
        if self.slab is not None and not self.slab.intact():
            raise RuntimeError("a dense parameter was re-allocated; build a new ShardedTrainer")
        if self._shape != (tuple(xb.shape), tuple(yb.shape)):
            self._build(xb, yb)
            self._pre = None
        lay, B = self.layout, xb.shape[0]
        cuda = xb.device.type == "cuda"
        if self.exchange == "direct" and cuda and self.slab is not None:
            return self._train_step_direct(xb, yb, next_xb)
        queues = cuda and self.slab is not None and os.environ.get("DCTR_SHARDED_QUEUES", "0") == "1"
        if not queues:
            return self._train_step_serial(xb, yb, next_xb)
        from ._hip import streams as _streams
        main = torch.cuda.current_stream(xb.device)
        if self._side is None:
            self._side = _streams.side_stream(xb.device, "shard")
            self._copy = _streams.side_stream(xb.device, "stage")
            # (events that live as long as the trainer: one created per step could be collected during a later hipGraph
            # capture, which HIP answers with an abort -- _hip/graph.py no_gc_during_capture)
            self._ev = {k: torch.cuda.Event() for k in ("copied", "computed", "dense", "loss")}
            self._side_busy = False
        side, cp, ev = self._side, self._copy, self._ev
        key = (xb.data_ptr(), xb._version)
        announced = self._announced == key and self._pre is not None
        if next_xb is not None and tuple(next_xb.shape) == tuple(xb.shape):
            next_key = (next_xb.data_ptr(), next_xb._version)
        else:
            next_key = None

        def copies():
            # what the captured compute segment reads.  (Behind the previous step's compute segment, which read the same
            # buffers; whatever produced xb / yb / next_xb ran on the caller's stream.)
            cp.wait_stream(main)
            with torch.cuda.stream(cp):
                self._x.copy_(xb, non_blocking=True)
                self._y.copy_(yb, non_blocking=True)
                if next_key is not None:
                    self._ids_next.copy_(self.ops.pack_ids(next_xb))
                ev["copied"].record(cp)

        # The HOST issues every piece of this step and is barely ahead of the GPU: what lies on the step's critical chain
        # is enqueued first, everything else behind it (round 4: with the side queue's work issued in front of the gradient
        # all-to-all the main queue idled ~50 us per step waiting for the host).
        if announced:
            chunks, self._ids_t = self._pre              # gathered at the end of the previous call
            self._pre = None
            dist.all_to_all_single(self._recv, chunks, group=self.group)             # rows -> samples' ranks
            copies()
        else:
            copies()
            main.wait_event(ev["copied"])
            self._ids_tmp.copy_(self.ops.pack_ids(self._x))
            ids_all = torch.empty_like(self._ids_tmp)
            dist.all_to_all_single(ids_all, self._ids_tmp, group=self.group)
            self._ids_view.copy_(ids_all.view(lay.world * B, lay.n_ids))
            chunks, self._ids_t = self._segB()
            self._pre = None
            dist.all_to_all_single(self._recv, chunks, group=self.group)             # rows -> samples' ranks
        main.wait_event(ev["copied"])
        if self._side_busy:
            main.wait_event(ev["dense"])          # the previous step's dense optimizer step wrote the weights read next
        self.overlap_wgrad = True
        send, loss, y_pred = self._segC()
        ev["computed"].record(main)
        wgrad = self.slab.deferred
        dist.all_to_all_single(self._grads_all, send, group=self.group)              # row gradients (+ next ids)
        self._segD()                                                                  # owners' update
        self._announced = next_key
        if next_key is not None:
            self._pre = self._segB()                                                  # owners' gather for the next call
        # ---- side queue: weight gradients, all-reduce, dense step -- beside the exchange and the update above
        side.wait_event(ev["computed"])
        with torch.cuda.stream(side):
            if wgrad is not None:
                wgrad(side)
            ev["loss"].record(side)                   # `loss` (finished by the reduction) is complete here
            work = dist.all_reduce(self.slab.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()
            self._segE()
            ev["dense"].record(side)
        self._side_busy = True
        main.wait_event(ev["loss"])                   # what this call returns is complete on the caller's stream
        return loss, loss.reshape(1), y_pred

    # ---- the whole step as ONE hipGraph over the direct exchange ----------------------------------------------------
    def _direct_setup(self, xb):
        lay, B, dev = self.layout, xb.shape[0], xb.device
        self._dx = DirectExchange(self.group, self.world, self.rank, dev, B, lay.ldc, lay.n_ids, self.slab.grad.numel(),
                                  dense_src=self.slab.grad)
        sub = self.ops.sub
        nu = sub.n_vcols if sub is not None else 1      # (one row of ids per X column feeding a unit)
        NB = lay.world * B
        self._chunks = torch.zeros((NB, lay.ldc), dtype=torch.float32, device=dev)
        self._ids_buf = torch.zeros((nu, NB), dtype=torch.int32, device=dev)
        self._parts_buf = torch.zeros((nu, NB), dtype=torch.int16, device=dev)
        self._send_buf = torch.zeros((NB, lay.ldc), dtype=torch.float32, device=dev)
        self._id_cols = torch.tensor(lay.id_cols, dtype=torch.int32, device=dev)
        self._direct_seg = _Segment(self._direct_body, bool(self.use_graphs))
        from ._hip import streams as _streams
        self._side = _streams.side_stream(dev, "shard")
        self._eng = self._engine_setup(B, dev)

    # ---- the compute segment as C-ABI calls: the tower launch reads its rows straight from the exchange buffer --------------
    def _engine_setup(self, B, dev):
        """Round 6.  The compute segment used to run through the model's autograd Functions: ``dctr_shard_assemble_fwd``
        copied the rows out of the receive buffer into the tower's input (+ the linear logit and the FM term), the tower launch
        read that copy, autograd delivered the three gradients to taps.  For a model whose logit is ``linear [+ FM] + tower``
        (``_gather_step``: DeepFM, WDL) the fused gather + tower launch of the single-GPU step engine
        (``dctr_embed_tower_train_step``, csrc/mlp.hip) does all of that itself when it is handed a plan whose TABLES ARE THE
        RECEIVE BUFFER: "table" of field f = the slot of f's owner in ``recv`` (rows ``ldc`` floats apart, one row per sample),
        "id" = the sample's own index (an extra column of the staged batch), the wide "tables" = the owners' summed wide
        values.  One launch and 7 us less per step (profiles/r06_sharded_1rank_timeline*.txt), no autograd graph, same
        arithmetic in the same order (tests/test_gpu_direct_exchange.py compare with the single-process step).
        -> dict of what the segment needs, or None (autograd route)."""
        import ctypes
        model, plan, lay, slab = self.model, self.plan, self.layout, self.slab
        if not getattr(model, "_gather_step", False) or not self._dx.push or self.ops.__class__ is not HipShardOps or \
                os.environ.get("DCTR_SHARDED_ENGINE", "1") == "0" or not plan.simple_units:
            return None
        from ._hip import lib as L
        from ._hip.plan import EmbeddingPlan
        from ._hip.step import GatherStep
        from .inputs import DenseFeat, SparseFeat
        eng = GatherStep(model, slab)
        b = eng._buffers(B, dev)
        if b is None:
            return None
        C = int(plan.n_xcols)
        recv = self._dx.recv                        # [world, B, ldc]: rank q's rows of MY samples arrive in recv[q]

        class _Tbl(object):
            def __init__(self, w):
                self.weight = w
        fi, deep_cols, wide_cols, deep_t, wide_t = {}, [], [], {}, {}
        for i in range(lay.F):
            q, sl = lay.owner[i], lay.slot[i]
            name = "recv_slot_%d" % i
            deep_cols.append(SparseFeat(name, B, lay.D))
            deep_t[name] = _Tbl(recv[q][:, sl * lay.D:(sl + 1) * lay.D])
            fi[name] = (C, C + 1)
        if lay.has_wide:
            for q in sorted(set(lay.owner)):
                name = "recv_wide_%d" % q
                wide_cols.append(SparseFeat(name, B, 1))
                wide_t[name] = _Tbl(recv[q][:, lay.wide_col:lay.wide_col + 1])
                fi[name] = (C, C + 1)
        for fc in model.dnn_feature_columns:
            if isinstance(fc, DenseFeat):
                deep_cols.append(fc)
                fi[fc.name] = model.feature_index[fc.name]
        for fc in model._linear_feature_columns:
            if isinstance(fc, DenseFeat):
                wide_cols.append(fc)
                fi[fc.name] = model.feature_index[fc.name]
        pseudo = EmbeddingPlan(fi, deep_columns=deep_cols, deep_tables=deep_t, wide_columns=wide_cols,
                               wide_tables=wide_t if wide_t else None, wide_dense_weight=plan.wide_dense_weight)
        if pseudo.width != plan.width or pseudo.ld_out != plan.ld_out or (pseudo.wide_dense_weight is None) != \
                (plan.wide_dense_weight is None):
            return None
        cp = pseudo.bind(dev)
        if L.lib().dctr_embed_tower_train_supported(cp, ctypes.byref(b.desc), int(B)) != 1:
            return None
        # the staged batch with one more column: the sample's own index, the "id" of every pseudo field
        xfull = torch.zeros((B, C + 1), dtype=torch.float32, device=dev)
        xfull[:, C] = torch.arange(B, dtype=torch.float32, device=dev)
        self._x = xfull[:, :C]
        return {"eng": eng, "b": b, "pseudo": pseudo, "xfull": xfull, "want_fm": eng.want_fm}

    def _compute_engine(self, xf=None, y=None, next_x=None, carry=True):
        """[rows in recv] -> fused gather + tower + head + BCE + backward-data (ONE launch) -> gradient assembly into the
        owners' buffers; the tower's weight gradients are left to the caller (``slab.deferred``), like TowerHeadFunction does.
        ``xf`` [B, C + 1] / ``y``: the batch in place (a step of a captured block: nothing was staged), its last column the
        sample index; default: the staged copy.  ``next_x``: the announced next batch, read in place by the gradient
        assembly."""
        import ctypes
        from ._hip import lib as L
        E = self._eng
        model, plan, slab, lay = self.model, self.plan, self.slab, self.layout
        b, pseudo = E["b"], E["pseudo"]
        xf = E["xfull"] if xf is None else xf
        dev = xf.device
        B = xf.shape[0]
        lib = L.lib()
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())       # noqa: E731
        cp = pseudo.bind(dev)
        y = (self._y if y is None else y).reshape(-1)
        if getattr(self, "_pending_join", False):
            # the previous step's weight gradients / dense sum / optimizer step (second queue) were left unjoined: the tower
            # launch below is the first reader of what they write -- joined HERE, behind the rows exchange, not at the end of
            # that step (the cross-queue edge then has the staging / exchange launches to hide behind)
            torch.cuda.current_stream(dev).wait_stream(self._side)
            self._pending_join = False
        y_pred = torch.empty((B,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ld = plan.ld_out
        ld_s = b.fm_s.stride(0) if b.fm_s is not None else 0
        bias = model.out.bias
        sh = L.stream_handle(dev)
        L.check(lib.dctr_embed_tower_train_step(cp, P(xf), xf.stride(0), ctypes.byref(b.desc), B, 1 if E["want_fm"] else 0,
                                                P(bias), P(y), P(y_pred), P(b.g_logit), P(b.gx), ld, P(b.out), ld, P(b.fm_s),
                                                ld_s, P(plan.err_flag(dev)), P(b.ws), sh), "dctr_embed_tower_train_step(recv)")
        g_bias = slab.grad_of(bias)

        def wgrad(stream):
            with torch.cuda.stream(stream):
                L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(b.desc), P(b.out), ld, B, P(b.g_logit), P(b.ws), P(loss),
                                                 P(g_bias), None, L.stream_handle(dev)), "dctr_mlp_train_wgrad")
        slab.deferred = wgrad
        g_wd = slab.grad_of(plan.wide_dense_weight) if plan.wide_dense_weight is not None else None
        send = self.ops.assemble_bwd(xf[:, :plan.n_xcols], b.gx, b.g_logit if plan.has_wide else None,
                                     b.g_logit if E["want_fm"] else None, b.out, b.fm_s,
                                     g_wd if plan.has_wide else None, send=self._send_static,
                                     push=self._dx.grad_tbl if self._dx.push else None, next_x=next_x, carry=carry)
        return send, loss, y_pred

    def _push_rows(self):
        return (self._dx.row_tbl, self._dx.recv.shape[1]) if self._dx.push else None

    def _direct_body(self, inplace=None, defer_join=False):
        """(``inplace`` = (xf [B, C + 1], y, next_x | None, announce): a step of a captured block on the engine compute
        segment -- the batch is read where it lies, nothing was staged; ``defer_join``: the second queue is joined in front of
        the NEXT step's tower launch instead of at the end of this step.)
        rows exchange -> [assemble, tower + head + backward-data, assemble^T] -> gradient exchange -> owners' update ->
        owners' gather for the next batch -> dense exchange + sum + optimizer step; the tower's weight gradients on a second
        queue beside everything behind the tower.  Every launch of a step, exchanges included, capturable: one hipGraph."""
        dx, lay, slab = self._dx, self.layout, self.slab
        main = torch.cuda.current_stream(self._x.device)
        side = self._side
        # the id-only half of the owners' update (every partition's entries found and sorted: dctr_embed_segments) on the
        # pre-pass queue, beside the rows exchange, the assembly and the tower: the ids of this batch have been sitting in
        # _ids_buf since the previous step's gather.  The update then runs its lean pre-sorted kernel.  OPT-IN
        # (DCTR_SHARDED_SEGMENTS=1): measured at one rank the step is SLOWER with it, 0.158 against 0.144 ms -- the third
        # queue inside the captured step costs more in cross-queue edges than the update kernel saves
        # (gpurun r4_s2_6, profiles/r04_bench_sharded_1rank_block.json).
        seg = None
        sub = self.ops.sub
        if sub is not None and os.environ.get("DCTR_SHARDED_SEGMENTS", "0") == "1" and sub.segments_enabled():
            seg = sub.launch_segments(self._ids_buf, self._parts_buf, self._ids_buf.shape[1])
        self._recv = dx.send_rows(self._chunks)
        self.overlap_wgrad = True
        self._send_static = self._send_buf
        try:
            if inplace is not None:
                send, loss, y_pred = self._compute_engine(inplace[0], inplace[1], inplace[2], carry=inplace[3])
            else:
                send, loss, y_pred = self._compute_engine() if getattr(self, "_eng", None) is not None else self._compute()
        finally:
            self._send_static = None
        wgrad = slab.deferred
        mode = self.state["mode"]

        def dense():
            if slab.begin_inline_step(mode[0], mode[1], mode[2] if len(mode) > 2 else 0.0):
                try:
                    dx.allreduce_dense(slab.grad, slab.inline)     # the sum kernel steps the parameters
                    slab.inline_done = True
                finally:
                    slab.end_inline_step()
                slab.step(*mode)                                   # (clears the flag)
            else:
                dx.allreduce_dense(slab.grad)
                slab.step(*mode)

        if wgrad is not None:
            # second queue: weight gradients -> dense exchange + sum + optimizer step, beside the gradient exchange, the
            # owners' update and their gather for the next batch (which touch tables only)
            side.wait_stream(main)
            wgrad(side)                              # (loss is finished by its reduction)
            with torch.cuda.stream(side):
                dense()
        grads_all = dx.send_grads(send)
        self.ops.update(grads_all, (self._ids_buf, self._parts_buf, seg))
        # (the ids of the announced next batch arrived with the gradients; an un-announced call gathers again itself)
        self.ops.gather(grads_all[:, lay.ids_col:lay.ids_col + lay.n_ids], out=(self._chunks, self._ids_buf, self._parts_buf),
                        push=self._push_rows())
        if wgrad is not None:
            if defer_join and getattr(self, "_eng", None) is not None:
                self._pending_join = True
            else:
                main.wait_stream(side)
        else:
            dense()
        return loss, y_pred

    def _stage(self, xb, yb, next_xb):
        """One launch in front of a direct-exchange step: the batch into the static buffers the step reads, the announced
        next batch's id columns into the gradient chunks (dctr_shard_stage); plain copies for other dtypes / strides."""
        import ctypes
        from ._hip import lib as L
        lay, B = self.layout, xb.shape[0]
        announce = next_xb is not None
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())       # noqa: E731
        yv = yb.reshape(-1)
        if xb.dtype == torch.float32 and xb.stride(1) == 1 and yv.dtype == torch.float32 and yv.is_contiguous() and \
                self._y.is_contiguous() and self._y.numel() == B and (not announce or next_xb.dtype == torch.float32):
            L.check(L.lib().dctr_shard_stage(P(xb), xb.stride(0), P(yv), B, xb.shape[1], P(self._x), self._x.stride(0),
                                             P(self._y), P(next_xb) if announce else None,
                                             next_xb.stride(0) if announce else 0, P(self._id_cols), lay.world, lay.n_ids,
                                             P(self._send_buf), lay.ldc, lay.ids_col, L.stream_handle(xb.device)),
                    "dctr_shard_stage")
        else:
            self._x.copy_(xb)
            self._y.copy_(yb)
            if announce:
                self._send_buf.view(lay.world, B, lay.ldc)[:, :, lay.ids_col:lay.ids_col + lay.n_ids].copy_(
                    self.ops.pack_ids(next_xb))

    # ---- S consecutive steps of the direct exchange as ONE hipGraph ---------------------------------------------------
    def train_block(self, x_block, y_block, next_first=None):
        """``x_block [S, B, C]`` / ``y_block [S, B]``: S consecutive train steps, equal to S ``train_step`` calls that each
        announce their successor.  With the direct exchange on a GPU the S steps -- staging, exchanges, tower, owners'
        update and gather, dense sum -- are ONE hipGraph replay on static block buffers (a step-per-graph replay leaves
        the GPU idle ~20 us between two launches: profiles/r04_sharded_1rank_direct_push_timeline.txt).
        ``next_first [B, C]``: the first batch of the NEXT call (its ids travel with the last step's gradients); every
        rank must announce (or not) consistently.  Returns the last step's ``(loss, total, y_pred)``."""
        S, B = int(x_block.shape[0]), int(x_block.shape[1])
        if not (self.exchange == "direct" and x_block.is_cuda and self.slab is not None):
            out, acc = None, None
            for j in range(S):
                nxt = x_block[j + 1] if j + 1 < S else next_first
                out = self.train_step(x_block[j], y_block[j], next_xb=nxt)
                t = out[0].detach().double().sum()
                acc = t if acc is None else acc + t
            self.last_block_loss = acc
            return out
        if not self.slab.intact():
            raise RuntimeError("a dense parameter was re-allocated; build a new ShardedTrainer")
        xb0, yb0 = x_block[0], y_block[0]
        if self._shape != (tuple(xb0.shape), tuple(yb0.shape)):
            self._build(xb0, yb0)
            self._pre = None
        if self._dx is None or self._dx.recv.shape[1] != B:
            self._direct_setup(xb0)
            self._announced = None
        blk = getattr(self, "_blk", None)
        C = int(x_block.shape[2])
        inplace = getattr(self, "_eng", None) is not None and x_block.dtype == torch.float32 and \
            y_block.dtype == torch.float32 and yb0.numel() == B and os.environ.get("DCTR_SHARDED_INPLACE", "1") != "0"
        if blk is None or tuple(blk["x"].shape) != tuple(x_block.shape) or blk["x"].dtype != x_block.dtype or \
                blk["dx"] is not self._dx or blk["inplace"] != inplace:
            # (engine compute segment: the block keeps one more column per row, the sample's own index -- the "id" the
            # tower launch looks its rows up with in the receive buffer -- and the steps read their batch where it lies)
            xfull = torch.zeros((S, B, C + 1), dtype=x_block.dtype, device=x_block.device)
            xfull[:, :, C] = torch.arange(B, dtype=torch.float32, device=x_block.device)[None, :]
            nffull = torch.zeros((B, C + 1), dtype=x_block.dtype, device=x_block.device)
            blk = self._blk = {"xfull": xfull, "x": xfull[:, :, :C],
                               "y": torch.empty((S,) + tuple(yb0.shape), dtype=y_block.dtype, device=y_block.device),
                               "nf": nffull[:, :C], "seg": {}, "dx": self._dx, "inplace": inplace}
        blk["x"].copy_(x_block)
        blk["y"].copy_(y_block.reshape(blk["y"].shape))
        announce = next_first is not None and tuple(next_first.shape) == tuple(xb0.shape)
        if announce:
            blk["nf"].copy_(next_first)
        key = (x_block.data_ptr(), x_block._version)
        if self._announced != key:                       # ids not at their owners yet: the explicit exchange + gather
            self._ids_tmp.copy_(self.ops.pack_ids(blk["x"][0]))
            ids_all = self._dx.send_ids(self._ids_tmp)
            self.ops.gather(ids_all, out=(self._chunks, self._ids_buf, self._parts_buf), push=self._push_rows())
        self._announced = (next_first.data_ptr(), next_first._version) if announce else None
        seg = blk["seg"].get((announce, bool(self.use_graphs)))
        if seg is None:
            sums = blk.setdefault("sums", {})
            skey = (announce, bool(self.use_graphs))

            def steps(blk=blk, announce=announce, S=S, sums=sums, skey=skey):
                out, losses = None, []
                for j in range(S):
                    nxt = blk["x"][j + 1] if j + 1 < S else (blk["nf"] if announce else None)
                    if blk["inplace"]:
                        out = self._direct_body(inplace=(blk["xfull"][j], blk["y"][j], nxt, nxt is not None),
                                                defer_join=j + 1 < S)
                    else:
                        self._stage(blk["x"][j], blk["y"][j], nxt)
                        out = self._direct_body()
                    losses.append(out[0].reshape(()))
                # (fit() logs the loss of EVERY step: the block's sum, in fp64, as part of the same graph)
                # (one tensor per captured variant: a replay refreshes the tensor of ITS capture)
                sums[skey] = torch.stack(losses).double().sum()
                return out
            seg = blk["seg"][(announce, bool(self.use_graphs))] = _Segment(steps, bool(self.use_graphs))
            # (the first call of a _Segment runs eagerly -- descriptor uploads, lazy buffers; the single-step segment has
            # normally done that already)
            seg.primed = getattr(self, "_direct_seg", None) is not None and self._direct_seg.primed
        loss, y_pred = seg()
        # fp64 scalar: this rank's data loss summed over the block's S steps
        self.last_block_loss = blk.get("sums", {}).get((announce, bool(self.use_graphs)))
        self._direct_watchdog(S)
        return loss, loss.reshape(1), y_pred

    def _direct_watchdog(self, n_steps):
        """A direct-exchange wait that timed out only raises a bit on the device and the step goes on with incomplete
        peer buffers: poll it every few hundred steps (one device sync), not only in close() -- a dead or desynchronised
        rank must not train silently wrong for the rest of the run (round-4 advisor finding)."""
        self._dx_steps = getattr(self, "_dx_steps", 0) + int(n_steps)
        if self._dx_steps >= int(os.environ.get("DCTR_DIRECT_CHECK_EVERY", "512")):
            self._dx_steps = 0
            if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
                self._dx.check()

    def _train_step_direct(self, xb, yb, next_xb=None):
        import ctypes
        from ._hip import lib as L
        lay, B = self.layout, xb.shape[0]
        if self._dx is None or self._dx.recv.shape[1] != B:
            self._direct_setup(xb)
            self._announced = None
        announce = next_xb is not None and tuple(next_xb.shape) == tuple(xb.shape) and next_xb.stride(1) == 1
        self._stage(xb, yb, next_xb if announce else None)
        key = (xb.data_ptr(), xb._version)
        if self._announced != key:                       # ids not at their owners yet: the explicit exchange + gather
            self._ids_tmp.copy_(self.ops.pack_ids(self._x))
            ids_all = self._dx.send_ids(self._ids_tmp)
            self.ops.gather(ids_all, out=(self._chunks, self._ids_buf, self._parts_buf), push=self._push_rows())
        self._announced = (next_xb.data_ptr(), next_xb._version) if announce else None
        loss, y_pred = self._direct_seg()
        self._direct_watchdog(1)
        return loss, loss.reshape(1), y_pred

    def _join(self):
        """The caller's stream waits for everything a previous train_step left on the trainer's own queues."""
        if self._side is not None and getattr(self, "_side_busy", False):
            torch.cuda.current_stream(self._side.device).wait_event(self._ev["dense"])
            self._side_busy = False

    def _train_step_serial(self, xb, yb, next_xb=None):
        """The same step on ONE queue in program order (CPU stand-ins over gloo, the autograd route,
        DCTR_SHARDED_QUEUES=0)."""
        lay, B = self.layout, xb.shape[0]
        self._x.copy_(xb)
        self._y.copy_(yb)
        key = (xb.data_ptr(), xb._version)
        if self._announced != key:                       # ids not here yet: the explicit exchange
            self._ids_tmp.copy_(self.ops.pack_ids(self._x))
            ids_all = torch.empty_like(self._ids_tmp)
            dist.all_to_all_single(ids_all, self._ids_tmp, group=self.group)
            self._ids_view.copy_(ids_all.view(lay.world * B, lay.n_ids))
        if next_xb is not None and tuple(next_xb.shape) == tuple(xb.shape):
            self._ids_next.copy_(self.ops.pack_ids(next_xb))
            self._announced = (next_xb.data_ptr(), next_xb._version)
        else:
            self._announced = None
        chunks, self._ids_t = self._segB()
        dist.all_to_all_single(self._recv, chunks, group=self.group)                 # rows -> samples' ranks
        send, loss, y_pred = self._segC()
        # (of the capture, when the segment is a graph replay)
        wgrad = self.slab.deferred if (self.slab is not None and xb.device.type == "cuda") else None
        if wgrad is not None:
            main = torch.cuda.current_stream(xb.device)
            if self._side is None:
                from ._hip import streams as _streams
                self._side = _streams.side_stream(xb.device, "shard")
            self._side.wait_stream(main)
            wgrad(self._side)
        dist.all_to_all_single(self._grads_all, send, group=self.group)              # row gradients (+ next ids)
        if wgrad is not None:
            main.wait_stream(self._side)                                              # the dense gradients are complete
        flat = self.slab.grad if self.slab is not None else self.bucket.flat
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._segD()                                                                  # overlaps with the all-reduce
        work.wait()
        self._segE()
        total = getattr(self, "_autograd_total", None) if self.slab is None else None
        return loss, (total if total is not None else loss.reshape(1)), y_pred

    def gather_tables(self):
        """Make every rank's copy of every table (and its optimizer state) current: owner -> all."""
        plan = self.plan
        from ._hip.plan import _STATE
        self._join()
        with torch.no_grad():
            for g, grp in enumerate(self.layout.groups):
                owner = self.layout.group_owner[g]
                params, seen = [], set()
                for f in [plan.deep[i] for i in grp["deep"]] + [plan.wide[k] for k in grp["wide"]]:
                    if id(f.param) not in seen:
                        seen.add(id(f.param))
                        params.append(f.param)
                for p_ in params:
                    for t in (p_.data, _STATE.get(p_)):
                        if t is None:
                            continue
                        _broadcast(t, owner, self.group)
