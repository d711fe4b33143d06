"""autograd.Function wrappers over the C-ABI (include/dctr.h).

Embedding tables never enter autograd as inputs: the reference's dense ``[V, D]`` gradient
(``aten::embedding_dense_backward``, triggered from basemodel.py:261) is replaced by an O(batch)
scatter executed inside ``EmbedFunction.backward``.  What that scatter does is selected by
``plan.update``:

  ``("dense",)``            add the row gradients into the table's zero-at-rest ``gacc`` slab and expose
                            it as ``param.grad`` -- bit-for-bit the tensor the reference hands to ANY
                            optimizer / regulariser (default; O(V) only if the optimizer is).
  ``("sgd", lr)``           ``table[row] -= lr * g`` straight from the scatter kernel.
  ``("sgd2", lr)``          two-pass SGD (needed when a max-pooled field re-reads the table).
  ``("adagrad", lr, eps)``  scatter into ``gacc``, then ``dctr_embed_apply`` consumes the touched rows.
"""
import contextlib
import ctypes
import os

import torch

from . import lib as L
from .plan import EmbeddingPlan


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _rows_f32(t, what):
    """2-D float32 tensor with unit inner stride (a row-strided view is fine)."""
    L.require_gpu(t, what)
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _aligned_rows(t, vec, ld_min):
    """Rows usable with ``vec``-wide loads: base pointer and row stride multiples of ``vec`` floats."""
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % vec or t.data_ptr() % (4 * vec) or t.stride(0) < ld_min:
        ld = (max(t.shape[1], ld_min) + 3) // 4 * 4
        buf = torch.empty((t.shape[0], ld), dtype=torch.float32, device=t.device)
        buf[:, :t.shape[1]].copy_(t)
        return buf, ld
    return t, t.stride(0)


class EmbedFunction(torch.autograd.Function):
    """Fused lookup: see ``dctr_embed_fwd`` / ``dctr_embed_bwd`` in include/dctr.h."""

    @staticmethod
    def forward(ctx, plan, X, anchor, wdense_w, want_fm, for_backward=False):
        lib = L.lib()
        X = _rows_f32(X, "model input X")
        if X.shape[1] < plan.n_xcols:
            raise ValueError("X has %d columns, the feature columns need %d" % (X.shape[1], plan.n_xcols))
        B = X.shape[0]
        cplan = plan.bind(X.device)
        out = torch.empty((B, plan.ld_out), dtype=torch.float32, device=X.device) if plan.has_lookup else None
        wide = torch.empty((B,), dtype=torch.float32, device=X.device) if plan.has_wide else None
        fm = torch.empty((B,), dtype=torch.float32, device=X.device) if want_fm else None
        if want_fm and (plan.emb_dim <= 0 or not plan.deep):
            raise ValueError("FM needs sparse features that share one embedding_dim")
        # exact lazy regularised / Adam update (csrc/lazy.hip): a training gather first replays the batch's rows to
        # the current step; any other reader of the tables gets them flushed
        lazy = plan.lazy if plan.update[0] == "lazy" else None
        if lazy is not None:
            if for_backward and lazy.plan is plan:
                lazy.catchup(X)
            else:
                lazy.flush()
        # side outputs for the deterministic fused update (only when a backward can follow)
        ids_t = parts_t = fm_s = den_t = amax = None
        ld_s = 0
        if for_backward and plan.table_params and plan.update_kernel_ok(B):
            ids_t = torch.empty((plan.n_vcols, B), dtype=torch.int32, device=X.device)
            # each entry's partition of the update kernel (a 16-bit tag: its workgroups compare instead of dividing)
            parts_t = torch.empty((plan.n_vcols, B), dtype=torch.int16, device=X.device)
            if want_fm:
                ld_s = (plan.emb_dim + 3) // 4 * 4
                fm_s = torch.empty((B, ld_s), dtype=torch.float32, device=X.device)
            # general units (pooled VarLen fields, shared tables): mean pooling's divisors (written with the ids) and
            # max pooling's arg-max positions (written by the gather) are this step's side buffers
            den_t, amax = plan.step_buffers(B, X.device)
        plan.point_step_buffers(den_t, amax)
        # The part of the update that needs only the ids -- finding and sorting every partition's entries -- runs on a
        # side stream, under the tower.  In the fused train step with in-kernel optimizer that stream also computes the
        # ids itself (from X, ahead of the gather) and later runs the update: its chain then waits for the main
        # stream once (for the tower's gradients) and the main chain -- gather, tower, weight gradients -- for nothing.
        sink = getattr(plan, "dense_sink", None)
        segs = ids_t is not None and plan.segments_enabled() and getattr(plan, "exchange", None) is None
        own_ids = segs and X.is_cuda and sink is not None and getattr(sink, "inline", None) is not None
        ctx.seg_event = None
        err = plan.err_flag(X.device)

        def gather(stream, with_ids, signal=None):
            # signal: the step's sync block (dense.DenseSlab.sync_block) -- the gather stores its outputs write-through
            # and signals DCTR_SYNC_GATHER; the tower's stream waits for that with dctr_step_wait, not for an event
            plan.cplan.step_sync = signal.data_ptr() if signal is not None else None
            fused_ids = with_ids and plan.gen is None      # (a general unit spans several X columns: dctr_embed_ids)
            try:
                L.check(lib.dctr_embed_fwd(cplan, _ptr(X), X.stride(0), B, _ptr(out), plan.ld_out, _ptr(wide), 1,
                                           _ptr(fm), _ptr(err), plan.units_ptr(), plan.n_grid_units,
                                           _ptr(ids_t) if fused_ids else None, _ptr(parts_t) if fused_ids else None,
                                           _ptr(fm_s), ld_s, stream), "dctr_embed_fwd")
            finally:
                plan.cplan.step_sync = None
            if with_ids and not fused_ids and ids_t is not None:
                L.check(lib.dctr_embed_ids(cplan, plan.units_ptr(), plan.n_grid_units, _ptr(X), X.stride(0), B,
                                           _ptr(ids_t), _ptr(parts_t), stream), "dctr_embed_ids")

        if own_ids and getattr(sink, "gather_side", False):
            # Topology "gather_side": the gather runs on the side stream too, in stream order behind the previous
            # step's update (the only thing it depends on), so the next step's rows are being fetched while the main
            # stream still finishes this step's weight gradients; the main stream waits for the gather alone.  Inside
            # a multi-step capture the side stream does not wait for the main one first (that would be the previous
            # step's weight-gradient reduction): it does only when it has to -- first step of a capture, eager steps
            # (whatever produced X ran on the main stream), tables last written elsewhere.
            sync = sink.sync_block(X.device) if getattr(sink, "flag_sync", False) else None
            ctx.seg_event = plan.launch_segments(ids_t, parts_t, B, before=lambda st: gather(st, True, sync),
                                                 fork=not sink.side_chain_open(plan._seg_stream),
                                                 join_before=sync is None)
            if ctx.seg_event[0] is not True:
                sink.update_stream = ctx.seg_event[0]
            if sync is not None:
                # Topology "flags": no graph edge between the two queues inside a step.  A one-wave kernel on THIS
                # stream waits for the gather's signal (4.6 us from the gather's last workgroup to the tower's first,
                # against 11-12 us through a cross-queue hipGraph edge: tools/micro/hopbench.hip)
                L.check(lib.dctr_step_wait(_ptr(sync), L.SYNC_GATHER, sink.sync_timeout_us,
                                           L.stream_handle(X.device)), "dctr_step_wait(gather)")
        else:
            if own_ids and getattr(sink, "wgrad_on_seg", False):
                # Topology "tower_seg": the side stream carries ids, pre-pass and (behind the tower) the weight gradients +
                # their reduction; the update runs on the MAIN stream.  Inside a multi-step capture the pre-pass does not
                # wait for the main stream (= for the previous step's update): with that edge hipGraph puts the weight
                # gradients on the gather's queue and serialises the step (tools/micro/topobench.hip, recipe V1b:
                # 127 us against 93).  What the edge protected is kept apart instead: the bucket workspace alternates
                # between two tensors, and the previous update's operands stay allocated until this step's update has
                # been enqueued (sink.upd_keep), so nothing the side stream writes now can be memory they still read.
                plan._ws_slot = 1 - getattr(plan, "_ws_slot", 1)
                ctx.seg_event = plan.launch_segments(ids_t, parts_t, B, X=X, slot=plan._ws_slot,
                                                     fork=not sink.side_chain_open(plan._seg_stream))
                if ctx.seg_event[0] is not True:
                    sink.update_stream = ctx.seg_event[0]
            elif own_ids:
                ctx.seg_event = plan.launch_segments(ids_t, parts_t, B, X=X)
                if ctx.seg_event[0] is not True:              # (True: the CPU stand-in, no streams)
                    sink.update_stream = ctx.seg_event[0]     # where this step's update will run (see backward)
            gather(L.stream_handle(X.device), not own_ids)
        ctx.plan, ctx.want_fm = plan, want_fm
        if segs and not own_ids:
            ctx.seg_event = plan.launch_segments(ids_t, parts_t, B)
        ctx.save_for_backward(X, out if (want_fm or plan.gen is not None) else None, ids_t, fm_s, parts_t, den_t, amax)
        ctx.set_materialize_grads(False)
        outs = (out if out is not None else X.new_zeros((B, 0)),
                wide if wide is not None else X.new_zeros((B,)),
                fm if fm is not None else X.new_zeros((B,)))
        return outs

    @staticmethod
    def backward(ctx, g_out, g_wide, g_fm):
        lib = L.lib()
        plan = ctx.plan
        X, out, ids_t, fm_s, parts_t, den_t, amax = ctx.saved_tensors
        B = X.shape[0]
        plan.point_step_buffers(den_t, amax)
        g_wd = None
        if not plan.has_lookup:
            g_out = None
        if not plan.has_wide:
            g_wide = None
        if not ctx.want_fm:
            g_fm = None
        g_w = None
        if g_wide is not None:
            g_wide = g_wide.contiguous()
            if plan.wide_dense_weight is not None and ctx.needs_input_grad[3]:
                # d wide / d Linear.weight = X_dense^T g  (basemodel.py:88-90): an extra workgroup of the
                # deterministic update kernel when that runs, else a GEMV
                if ids_t is not None and getattr(plan, "exchange", None) is None and plan.table_params:
                    sink = getattr(plan, "dense_sink", None)
                    g_wd = sink.grad_of(plan.wide_dense_weight) if sink is not None else None
                    if g_wd is None:
                        g_wd = torch.empty((len(plan.wdense_cols), 1), dtype=torch.float32, device=X.device)
                        g_w = g_wd
                else:
                    g_w = plan.dense_matrix(X, plan.wdense_cols).t().mv(g_wide).unsqueeze(1)
        if g_fm is not None:
            g_fm = g_fm.contiguous()
        ld_g = 0
        if g_out is not None:
            g_out, ld_g = _aligned_rows(g_out, plan.vec, 0)
        if (g_out is None and g_fm is None and g_wide is None) or not plan.table_params:
            return None, None, None, g_w, None, None

        if getattr(plan, "exchange", None) is not None:
            # data-parallel: the trainer all-gathers the row gradients and applies the global update
            plan.exchange(X=X, g_out=g_out, out=out, fm_s=fm_s, g_fm=g_fm, g_wide=g_wide, amax=amax)
            return None, None, None, g_w, None, None

        update = plan.update
        kind = update[0]
        stream = L.stream_handle(X.device)

        if kind == "lazy":
            lazy = plan.lazy
            if lazy is None or lazy.plan is not plan or ids_t is None:
                raise NotImplementedError("the lazy regularised / Adam table update needs lookups through the model's "
                                          "own plan and a batch the deterministic update kernel supports "
                                          "(DCTR_LAZY_UPDATE=0 selects the exact dense path)")
            lazy._ensure(X.device)
            cplan = plan.bind(X.device)
            ws, ws_n, pre = plan.update_workspace_for(ids_t, ctx.seg_event, B)
            # round 6: with pre-sorted entries the regularised / Adam step runs at the row, inside the sorted update (no
            # gradient slab, no second pass over the batch's rows: csrc/update_kernels.hpp DCTR_UPD_LAZY)
            if pre and lazy.update_fused(plan, cplan, ids_t, parts_t, B, g_out, ld_g, out, fm_s, g_fm, g_wide, X, g_wd, ws,
                                         ws_n):
                return None, None, None, g_w, None, None
            L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), plan.n_grid_units, plan.max_vocab, _ptr(ids_t),
                                          _ptr(parts_t), B, _ptr(g_out), ld_g, _ptr(out), plan.ld_out, _ptr(fm_s),
                                          fm_s.stride(0) if fm_s is not None else 0, _ptr(g_fm), _ptr(g_wide), 1,
                                          L.UPD_ACCUM, 0.0, 0.0, _ptr(X), X.stride(0), _ptr(g_wd), None, _ptr(ws), ws_n,
                                          pre, stream), "dctr_embed_update(accumulate)")
            lazy.apply(ids_t)
            return None, None, None, g_w, None, None

        if ids_t is not None:
            # deterministic single-pass path (csrc/update.hip): no atomics, optimizer fused in
            if kind == "dense":
                plan.ensure_gacc()
                plan.prepare_dense_grads()
                opt, lr, eps = L.UPD_ACCUM, 0.0, 0.0
            elif kind in ("sgd", "sgd2"):
                opt, lr, eps = L.UPD_SGD, float(update[1]), 0.0
            elif kind == "adagrad":
                opt, lr, eps = L.UPD_ADAGRAD, float(update[1]), float(update[2])
            else:
                raise RuntimeError("unknown sparse update mode %r" % (kind,))
            cplan = plan.bind(X.device)
            sink = getattr(plan, "dense_sink", None)
            # (armed AND carried out by the tower + head kernel of this step: only then has its event been recorded and
            # does nobody else step Linear.weight)
            inline = getattr(sink, "inline", None) if (sink is not None and getattr(sink, "inline_done", False)) else None
            side = None
            if inline is not None and X.is_cuda and ctx.seg_event is not None and ctx.seg_event[0] is not True and \
                    sink.update_stream is ctx.seg_event[0]:
                # fused train step with in-kernel optimizer: the update leaves the critical chain -- it runs on the
                # pre-pass's side stream, behind the pre-pass and behind the tower kernel that produced its gradients,
                # beside the tower's weight-gradient kernels (which stay on the main stream).  DenseSlab.join() brings
                # the streams together at the end of the step.
                side = ctx.seg_event[0]      # (already waiting for the tower + head launch: mlp.TowerHeadFunction)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                ws, ws_n, pre = plan.update_workspace_for(ids_t, ctx.seg_event, B)
                wd = ctypes.byref(inline) if (inline is not None and g_wd is not None and g_w is None) else None
                L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), plan.n_grid_units, plan.max_vocab, _ptr(ids_t),
                                              _ptr(parts_t), B, _ptr(g_out), ld_g, _ptr(out), plan.ld_out, _ptr(fm_s),
                                              fm_s.stride(0) if fm_s is not None else 0, _ptr(g_fm), _ptr(g_wide), 1,
                                              opt, lr, eps, _ptr(X), X.stride(0), _ptr(g_wd), wd, _ptr(ws), ws_n, pre,
                                              L.stream_handle(X.device)), "dctr_embed_update")
            if side is not None:
                # (everything the side-stream kernels touch stays allocated until the join)
                sink.forked(side, (X, out, ids_t, parts_t, fm_s, g_out, g_fm, g_wide, g_wd, ws, den_t, amax))
            elif sink is not None and getattr(sink, "wgrad_on_seg", False):
                sink.upd_keep = (X, out, ids_t, parts_t, fm_s, g_out, g_fm, g_wide, g_wd, ws, den_t, amax)   # (see forward)
            after = getattr(sink, "after_update", None) if sink is not None else None
            if after is not None:        # topology "tower_side": the weight gradients fork off behind the update's launch
                sink.after_update = None
                after()
            return None, None, None, g_w, None, None

        # general path (pooled VarLen fields, shared tables, very large batches): atomic scatter (+ consume pass)
        if kind == "sgd" and not plan.has_maxpool:
            cplan = plan.bind(X.device)
            L.check(lib.dctr_embed_bwd(cplan, _ptr(X), X.stride(0), B, _ptr(g_out), ld_g, _ptr(out), plan.ld_out,
                                       _ptr(g_fm), _ptr(g_wide), L.BWD_SGD, float(update[1]), stream),
                    "dctr_embed_bwd(sgd)")
            return None, None, None, g_w, None, None

        plan.ensure_gacc()
        if kind == "dense":
            plan.prepare_dense_grads()
        cplan = plan.bind(X.device)
        L.check(lib.dctr_embed_bwd(cplan, _ptr(X), X.stride(0), B, _ptr(g_out), ld_g, _ptr(out), plan.ld_out,
                                   _ptr(g_fm), _ptr(g_wide), L.BWD_ACCUM, 0.0, stream), "dctr_embed_bwd(accum)")
        if kind in ("sgd", "sgd2"):
            L.check(lib.dctr_embed_apply(cplan, _ptr(X), X.stride(0), B, L.OPT_SGD, float(update[1]), 0.0, stream),
                    "dctr_embed_apply(sgd)")
        elif kind == "adagrad":
            L.check(lib.dctr_embed_apply(cplan, _ptr(X), X.stride(0), B, L.OPT_ADAGRAD, float(update[1]),
                                         float(update[2]), stream), "dctr_embed_apply(adagrad)")
        elif kind != "dense":
            raise RuntimeError("unknown sparse update mode %r" % (kind,))
        return None, None, None, g_w, None, None


def embed(plan, X, want_fm=False, full=False):
    """(out [B, width] view, wide [B], fm [B]) for model input ``X`` under ``plan``.  ``full`` returns the
    un-sliced ``[B, ld_out]`` buffer (the MFMA tower reads the first ``plan.width`` columns of it and hands back
    a gradient of the same shape, so no slice / zero-fill kernels appear in the autograd graph)."""
    L.require_gpu(X, "model input X")
    sharder = getattr(plan, "sharder", None)
    if sharder is not None:          # table-sharded multi-GPU training (parallel.ShardedTrainer)
        return sharder.embed(X, want_fm, full)
    owner = getattr(plan, "_owner", None)
    if owner is not None and getattr(owner, "sharder", None) is not None and torch.is_grad_enabled():
        # A SECONDARY plan over tables that a ShardedTrainer has sharded (gather_columns / input_from_feature_columns /
        # Linear.forward next to the model's own fused lookup: IFM / DIFM-style models): this rank's copy of a table it
        # does not own is stale and the local update below would fork it -- training would diverge silently (round-3
        # advisor finding).  Only lookups through the model plan go through the exchange.
        raise NotImplementedError("this lookup reads tables that are sharded across ranks (ShardedTrainer) through a "
                                  "secondary plan: only the model's own fused lookup is routed through the exchange; "
                                  "train this model with DataParallelTrainer")
    plan.bind(X.device)
    out, wide, fm = EmbedFunction.apply(plan, X, plan.anchor, plan.wide_dense_weight, bool(want_fm),
                                        torch.is_grad_enabled())
    if plan.has_lookup and not full:
        out = out[:, :plan.width]
    return out, wide, fm


_PLAN_CACHE_ATTR = "_dctr_plans"


class SplitGatheredFunction(torch.autograd.Function):
    """``full [B, ld]`` (``embed(..., full=True)``) -> (``full[:, :W].view(B, F, D)``, ``full[:, W:W + nd]``): the two views every
    interaction model takes of the gather's output.  As plain slices their backward is three zero-fills, three copies and
    an add of [B, ld] tensors (7 launches, ~30 us at the Criteo shape); here it is two copies into one buffer."""

    @staticmethod
    def forward(ctx, full, W, F, D, nd):
        ctx.dims = (full.shape[0], full.shape[1], int(W), int(nd))
        emb = full[:, :W].view(full.shape[0], F, D)
        return emb, full[:, W:W + nd]

    @staticmethod
    def backward(ctx, g_emb, g_dense):
        B, ld, W, nd = ctx.dims
        ref = g_emb if g_emb is not None else g_dense
        g = torch.empty((B, ld), dtype=ref.dtype, device=ref.device)
        if ref.is_cuda and ref.dtype == torch.float32 and W % 4 == 0 and ld % 4 == 0 and \
                os.environ.get("DCTR_GLUE_KERNELS", "1") != "0":
            # one launch (csrc/head.hip k_rows_join) instead of two copies and a fill
            ge = g_emb.reshape(B, W) if g_emb is not None else None
            if ge is not None and (ge.stride(1) != 1 or ge.stride(0) % 4 or ge.data_ptr() % 16):
                ge = ge.contiguous()
            gd = g_dense if (g_dense is not None and nd > 0) else None
            if gd is not None and (gd.stride(1) != 1 or gd.dtype != torch.float32):
                gd = gd.float().contiguous()
            L.check(L.lib().dctr_rows_join(_ptr(ge), ge.stride(0) if ge is not None else 0, None, 0, W, _ptr(gd),
                                           gd.stride(0) if gd is not None else 0, nd if gd is not None else 0, _ptr(g), ld,
                                           B, L.stream_handle(ref.device)), "dctr_rows_join")
            return g, None, None, None, None
        if g_emb is not None:
            g[:, :W].copy_(g_emb.reshape(B, W))
        else:
            g[:, :W].zero_()
        if g_dense is not None and nd > 0:
            g[:, W:W + nd].copy_(g_dense)
            if ld > W + nd:
                g[:, W + nd:].zero_()
        elif ld > W:
            g[:, W:].zero_()
        return g, None, None, None, None


def split_gathered(full, plan):
    """(emb [B, F, D] view, dense [B, n_dense] view or None) of ``embed(plan, X, full=True)[0]``."""
    nd = len(plan.dense_cols)
    if not full.requires_grad or full.stride(1) != 1:
        emb = full[:, :plan.emb_width].reshape(full.shape[0], len(plan.deep), plan.emb_dim)
        return emb, (full[:, plan.emb_width:plan.emb_width + nd] if nd else None)
    emb, dense = SplitGatheredFunction.apply(full, plan.emb_width, len(plan.deep), plan.emb_dim, nd)
    return emb, (dense if nd else None)


def gather_columns(X, embedding_dict, feature_index, columns, pooled=True):
    """Per-column embeddings as views of ONE fused gather: ``[B, 1, D]`` per SparseFeat (and per pooled
    VarLenSparseFeat), ``[B, maxlen, D]`` per un-pooled VarLenSparseFeat.  Backs the reference-shaped
    helpers (``embedding_lookup``, ``varlen_embedding_lookup``, ``input_from_feature_columns``)."""
    cache = embedding_dict.__dict__.setdefault(_PLAN_CACHE_ATTR, {})
    key = (tuple(c.name for c in columns), bool(pooled), id(feature_index))
    plan = cache.get(key)
    if plan is None:
        plan = EmbeddingPlan(feature_index, deep_columns=list(columns), deep_tables=embedding_dict,
                             unpooled=not pooled, with_dense=False)
        owner = getattr(embedding_dict, "_dctr_owner_plan", None)
        if owner is not None:
            plan.share_update_with(owner)
        cache[key] = plan
    out, _, _ = embed(plan, X)
    B = X.shape[0]
    sparse_cols = [c for c in columns if not hasattr(c, "maxlen")]
    varlen_cols = [c for c in columns if hasattr(c, "maxlen")]
    views, off = {}, 0
    for c in sparse_cols:
        d = embedding_dict[c.embedding_name].weight.shape[1]
        views[c.name] = out[:, off:off + d].unsqueeze(1)
        off += d
    for c in varlen_cols:
        d = embedding_dict[c.embedding_name].weight.shape[1]
        t = 1 if pooled else c.maxlen
        views[c.name] = out[:, off:off + t * d].reshape(B, t, d)
        off += t * d
    return [views[c.name] for c in columns]


# ---- FM on explicit tensors (interaction.py:26-34) --------------------------------------------------
class FMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E):
        lib = L.lib()
        L.require_gpu(E, "FM input")
        if E.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % E.dim())
        B, F, D = E.shape
        if E.dtype != torch.float32 or E.stride(2) != 1 or E.stride(1) != D:
            E = E.float().contiguous()
        y = torch.empty((B,), dtype=torch.float32, device=E.device)
        L.check(lib.dctr_fm_fwd(_ptr(E), E.stride(0) if B > 1 else F * D, B, F, D, _ptr(y),
                                L.stream_handle(E.device)), "dctr_fm_fwd")
        ctx.save_for_backward(E)
        return y.unsqueeze(1)

    @staticmethod
    def backward(ctx, gy):
        lib = L.lib()
        (E,) = ctx.saved_tensors
        B, F, D = E.shape
        gy = gy.reshape(B).contiguous().float()
        gE = torch.empty((B, F, D), dtype=torch.float32, device=E.device)
        L.check(lib.dctr_fm_bwd(_ptr(E), E.stride(0) if B > 1 else F * D, B, F, D, _ptr(gy), _ptr(gE), F * D, 0,
                                L.stream_handle(E.device)), "dctr_fm_bwd")
        return gE


class InteractFunction(torch.autograd.Function):
    """InteractingLayer on ``E [B, F, D]`` (csrc/interact.hip): ``(E, Wq, Wk, Wv, Wr | None) -> [B, F, D]``."""

    @staticmethod
    def forward(ctx, E, Wq, Wk, Wv, Wr, heads, scaling):
        lib = L.lib()
        E, lde = _rows3(E, "InteractingLayer input")
        B, F, D = E.shape
        ws = [w.detach().float().contiguous() if w is not None else None for w in (Wq, Wk, Wv, Wr)]
        out = torch.empty((B, F, D), dtype=torch.float32, device=E.device)
        L.check(lib.dctr_interacting_fwd(_ptr(E), lde, B, F, D, int(heads), int(bool(scaling)), _ptr(ws[0]), _ptr(ws[1]),
                                         _ptr(ws[2]), _ptr(ws[3]), _ptr(out), F * D, L.stream_handle(E.device)),
                "dctr_interacting_fwd")
        ctx.save_for_backward(E, *[w for w in ws if w is not None])
        ctx.cfg = (int(heads), bool(scaling), ws[3] is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        heads, scaling, has_res = ctx.cfg
        saved = ctx.saved_tensors
        E, Wq, Wk, Wv = saved[0], saved[1], saved[2], saved[3]
        Wr = saved[4] if has_res else None
        E, lde = _rows3(E, "InteractingLayer input")
        B, F, D = E.shape
        dev = E.device
        gout = gout.reshape(B, F * D)
        if gout.dtype != torch.float32 or gout.stride(1) != 1:
            gout = gout.float().contiguous()
        gE = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        gWq, gWk, gWv = torch.empty_like(Wq), torch.empty_like(Wk), torch.empty_like(Wv)
        gWr = torch.empty_like(Wr) if has_res else None
        ws = torch.empty((max(1, lib.dctr_interacting_bwd_workspace_floats(B, D)),), dtype=torch.float32, device=dev)
        L.check(lib.dctr_interacting_bwd(_ptr(E), lde, B, F, D, heads, int(scaling), _ptr(Wq), _ptr(Wk), _ptr(Wv), _ptr(Wr),
                                         _ptr(gout), gout.stride(0), _ptr(gE), F * D, _ptr(gWq), _ptr(gWk), _ptr(gWv),
                                         _ptr(gWr), _ptr(ws), L.stream_handle(dev)), "dctr_interacting_bwd")
        return gE, gWq, gWk, gWv, gWr, None, None


def interacting_supported(F, D, H):
    return bool(L.lib().dctr_interacting_supported(int(F), int(D), int(H)))


class AFMFunction(torch.autograd.Function):
    """AFMLayer on ``E [B, F, D]`` (csrc/afm.hip): ``(E, W [D, A], b [A], h [A, 1], p [D, 1]) -> [B, 1]``."""

    @staticmethod
    def forward(ctx, E, W, b, h, p):
        lib = L.lib()
        E, lde = _rows3(E, "AFM input")
        B, F, D = E.shape
        A = W.shape[1]
        if D > 64 or A > 32 or F > 64 or F < 2:
            raise NotImplementedError("the gfx950 AFM kernel supports 2 <= fields <= 64, embedding_dim <= 64, "
                                      "attention_factor <= 32 (got F=%d, D=%d, A=%d)" % (F, D, A))
        Wc, bc, hc, pc = (t.detach().float().contiguous() for t in (W, b, h.reshape(-1), p.reshape(-1)))
        y = torch.empty((B,), dtype=torch.float32, device=E.device)
        L.check(lib.dctr_afm_fwd(_ptr(E), lde, B, F, D, A, _ptr(Wc), _ptr(bc), _ptr(hc), _ptr(pc), _ptr(y),
                                 L.stream_handle(E.device)), "dctr_afm_fwd")
        ctx.save_for_backward(E, Wc, bc, hc, pc)
        ctx.shapes = (tuple(h.shape), tuple(p.shape))
        return y.unsqueeze(1)

    @staticmethod
    def backward(ctx, gy):
        lib = L.lib()
        E, W, b, h, p = ctx.saved_tensors
        E, lde = _rows3(E, "AFM input")
        B, F, D = E.shape
        A = W.shape[1]
        dev = E.device
        gy = gy.reshape(B).contiguous().float()
        gE = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        gW, gb = torch.empty_like(W), torch.empty_like(b)
        gh, gp = torch.empty_like(h), torch.empty_like(p)
        ws = torch.empty((max(1, lib.dctr_afm_bwd_workspace_floats(B, D, A)),), dtype=torch.float32, device=dev)
        L.check(lib.dctr_afm_bwd(_ptr(E), lde, B, F, D, A, _ptr(W), _ptr(b), _ptr(h), _ptr(p), _ptr(gy), _ptr(gE), F * D,
                                 _ptr(gW), _ptr(gb), _ptr(gh), _ptr(gp), _ptr(ws), L.stream_handle(dev)), "dctr_afm_bwd")
        return gE, gW, gb, gh.reshape(ctx.shapes[0]), gp.reshape(ctx.shapes[1])


class BiPoolFunction(torch.autograd.Function):
    """BiInteractionPooling on the gather's rows (csrc/fm.hip): ``G [B, ld]`` (fields first, dense block at
    ``dense_off``) -> ``[B, r4(D + n_dense)]`` = ``[bi | dense]``, the NFM tower's input; the backward hands back a
    gradient with G's layout."""

    @staticmethod
    def forward(ctx, G, F, D, dense_off, n_dense):
        lib = L.lib()
        L.require_gpu(G, "BiInteractionPooling input")
        if G.dtype != torch.float32 or G.dim() != 2 or G.stride(1) != 1:
            G = G.float().contiguous()
        B = G.shape[0]
        ld_o = (D + n_dense + 3) // 4 * 4
        out = torch.zeros((B, ld_o), dtype=torch.float32, device=G.device) if ld_o != D + n_dense else \
            torch.empty((B, ld_o), dtype=torch.float32, device=G.device)
        L.check(lib.dctr_bi_pooling_fwd(_ptr(G), G.stride(0), B, F, D, dense_off, n_dense, _ptr(out), ld_o,
                                        L.stream_handle(G.device)), "dctr_bi_pooling_fwd")
        ctx.save_for_backward(G)
        ctx.dims = (F, D, dense_off, n_dense)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        (G,) = ctx.saved_tensors
        F, D, dense_off, n_dense = ctx.dims
        B = G.shape[0]
        if gout.dtype != torch.float32 or gout.stride(1) != 1:
            gout = gout.float().contiguous()
        gG = torch.zeros_like(G) if G.shape[1] != F * D + n_dense or (n_dense and dense_off != F * D) else \
            torch.empty_like(G)
        L.check(lib.dctr_bi_pooling_bwd(_ptr(G), G.stride(0), B, F, D, dense_off, n_dense, _ptr(gout), gout.stride(0),
                                        _ptr(gG), gG.stride(0), L.stream_handle(G.device)), "dctr_bi_pooling_bwd")
        return gG, None, None, None, None


# ---- CIN layer (interaction.py:207-248) ----------------------------------------------------------------
def _rows3(t, what):
    """[B, R, D] float32 with contiguous (R, D) rows; the batch stride may be anything >= R*D (views of the
    gather's output and of a previous layer's feature maps pass through without a copy)."""
    L.require_gpu(t, what)
    if t.dtype != torch.float32:
        t = t.float()
    B, R, D = t.shape
    if (D > 1 and t.stride(2) != 1) or (R > 1 and t.stride(1) != D) or (B > 1 and t.stride(0) < R * D):
        t = t.contiguous()
    return t, (t.stride(0) if B > 1 else R * D)


def cin_layer_forward(H, X0, W2d, bias, relu):
    """A = act(W . (H (x) X0) + bias): ``[B, h, D], [B, M, D] -> [B, O, D]`` (no autograd; see CINLayerFunction)."""
    lib = L.lib()
    H, ldh = _rows3(H, "CIN hidden input")
    X0, ldx = _rows3(X0, "CIN field input")
    B, h, D = H.shape
    M = X0.shape[1]
    O = W2d.shape[0]
    if M > 32:
        raise NotImplementedError("the gfx950 CIN kernels support at most 32 fields (got %d)" % M)
    W2d = W2d.contiguous()
    A = torch.empty((B, O, D), dtype=torch.float32, device=H.device)
    ws = torch.empty((lib.dctr_cin_workspace_floats(h, M, O),), dtype=torch.float32, device=H.device)
    L.check(lib.dctr_cin_layer_fwd(_ptr(H), ldh, _ptr(X0), ldx, _ptr(W2d), _ptr(bias), B, h, M, D, O, int(bool(relu)),
                                   _ptr(A), O * D, _ptr(ws), L.stream_handle(H.device)), "dctr_cin_layer_fwd")
    return A


class CINLayerFunction(torch.autograd.Function):
    """One CIN layer with the activation (relu or none) fused: ``dctr_cin_layer_fwd`` / ``dctr_cin_layer_bwd``."""

    @staticmethod
    def forward(ctx, H, X0, W2d, bias, relu):
        A = cin_layer_forward(H, X0, W2d, bias, relu)
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(H, X0, W2d, A if relu else None)
        return A

    @staticmethod
    def backward(ctx, gA):
        lib = L.lib()
        H, X0, W2d, A = ctx.saved_tensors
        H, ldh = _rows3(H, "CIN hidden input")
        X0, ldx = _rows3(X0, "CIN field input")
        B, h, D = H.shape
        M, O = X0.shape[1], W2d.shape[0]
        gA = gA.contiguous().float()
        W2d = W2d.contiguous()
        dev = H.device
        gH = torch.empty((B, h, D), dtype=torch.float32, device=dev)
        gX0 = torch.empty((B, M, D), dtype=torch.float32, device=dev)
        gW = torch.empty((O, h * M), dtype=torch.float32, device=dev)
        gb = torch.empty((O,), dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = torch.empty((max(1, lib.dctr_cin_bwd_workspace_floats(B, h, M, D, O)),), dtype=torch.float32, device=dev)
        L.check(lib.dctr_cin_layer_bwd(_ptr(gA), _ptr(A), O * D, int(ctx.relu), _ptr(H), ldh, _ptr(X0), ldx, _ptr(W2d),
                                       B, h, M, D, O, _ptr(gH), h * D, _ptr(gX0), M * D, 0, _ptr(gW), _ptr(gb),
                                       _ptr(ws), L.stream_handle(dev)), "dctr_cin_layer_bwd")
        return gH, gX0, gW, gb, None


class CINStackFunction(torch.autograd.Function):
    """The whole CIN (interaction.py:207-248) as ONE autograd node, optionally with the bias-free 1-unit projection
    xDeepFM puts on its output (xdeepfm.py:72, :97):

        x        [B, F, D] field matrix, or the gather's [B, >= F*D] row matrix whose first F*D columns are the fields
                 (xDeepFM: the gradient comes back in that shape -- no view backward (fill + slice copy) behind the CIN)
        wb       W_1, b_1, W_2, b_2, ...: the conv1ds' ``[O, h*F, 1]`` weights and ``[O]`` biases (None = no bias)
        w_head   None -> returns the CIN output ``[B, featuremap_num]``; ``[1, featuremap_num]`` -> returns
                 ``[B, 1]`` = output @ w_head.T

    Every layer's pooling kernel writes its block of the ``[B, featuremap_num]`` output in place (no torch.cat) and the
    backward reads the blocks' gradients in place (no slice copies); the layers' field-matrix gradients accumulate in one
    buffer inside the kernels (no adds); with ``w_head`` the projection's backward towards the layers is folded into the
    gradient-assembly kernel (dctr_cin_pool_bwd) and its forward is one wave-per-row dot product (dctr_rows_dot).
    Per layer: dctr_cin_layer_fwd + dctr_cin_pool_fwd forward, dctr_cin_pool_bwd + dctr_cin_layer_bwd backward."""

    @staticmethod
    def forward(ctx, x, F, D, relu, split_half, w_head, *wb):
        lib = L.lib()
        if x.dim() == 2:
            if x.shape[1] < F * D or x.stride(1) != 1 or x.dtype != torch.float32:
                raise ValueError("CIN: the row matrix must hold F*D contiguous float32 columns")
            X0 = x[:, :F * D].unflatten(1, (F, D))
        else:
            X0 = x
        B = X0.shape[0]
        dev = x.device
        n = len(wb) // 2
        sizes = [wb[2 * i].shape[0] for i in range(n)]
        geo = []                          # per layer: (O, n_hidden, pool_from, offset of its block)
        off = 0
        for i, O in enumerate(sizes):
            last = i == n - 1
            nh = 0 if last else (O // 2 if split_half else O)
            pf = nh if split_half else 0
            geo.append((O, nh, pf, off))
            off += O - pf
        fm = off
        feat = torch.empty((B, fm), dtype=torch.float32, device=dev)
        As = []
        hidden = X0
        for i in range(n):
            O, nh, pf, o0 = geo[i]
            bias = wb[2 * i + 1]
            A = cin_layer_forward(hidden, X0, wb[2 * i].reshape(O, -1), bias, relu)
            if B > 0:
                L.check(lib.dctr_cin_pool_fwd(_ptr(A), B, O, D, pf, ctypes.c_void_p(feat.data_ptr() + 4 * o0), fm,
                                              L.stream_handle(dev)), "dctr_cin_pool_fwd")
            As.append(A)
            hidden = A[:, :nh] if nh > 0 else None
        ctx.geo, ctx.relu, ctx.F, ctx.D, ctx.fm = geo, bool(relu), int(F), int(D), fm
        ctx.flat = x.dim() == 2
        ctx.x_cols = x.shape[1] if ctx.flat else 0
        ctx.has_head = w_head is not None
        ctx.has_bias = [wb[2 * i + 1] is not None for i in range(n)]
        ctx.w_shapes = [tuple(wb[2 * i].shape) for i in range(n)]
        ctx.save_for_backward(x, w_head, feat if ctx.has_head else None, *(list(wb[0::2]) + As))
        if not ctx.has_head:
            return feat
        if w_head.numel() != fm:
            raise ValueError("CIN head: %d weights for %d feature maps" % (w_head.numel(), fm))
        logit = torch.empty((B, 1), dtype=torch.float32, device=dev)
        wh = w_head.detach().contiguous()
        L.check(lib.dctr_rows_dot(_ptr(feat), fm, _ptr(wh), B, fm, _ptr(logit), L.stream_handle(dev)), "dctr_rows_dot")
        return logit

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        sv = ctx.saved_tensors
        x, w_head, feat = sv[0], sv[1], sv[2]
        n = len(ctx.geo)
        Ws, As = sv[3:3 + n], sv[3 + n:3 + 2 * n]
        F, D, fm = ctx.F, ctx.D, ctx.fm
        dev = x.device
        B = x.shape[0]
        st = L.stream_handle(dev)
        g = g.contiguous().float()
        ctx_join = False
        if ctx.flat:
            X0 = x[:, :F * D].unflatten(1, (F, D))
            gx = torch.empty((B, ctx.x_cols), dtype=torch.float32, device=dev)
            # (the dense columns behind the fields are not the CIN's inputs: zeroed by the joining launch at the end, or here)
            ctx_join = B > 0 and (F * D) % 4 == 0 and ctx.x_cols % 4 == 0 and os.environ.get("DCTR_GLUE_KERNELS", "1") != "0"
            if ctx.x_cols > F * D and not ctx_join:
                gx[:, F * D:].zero_()
            ld_gx = ctx.x_cols
        else:
            X0 = x
            gx = torch.empty((B, F, D), dtype=torch.float32, device=dev)
            ld_gx = F * D
        X0r, ldx = _rows3(X0, "CIN field input")
        g_wh = None
        if ctx.has_head:
            if B > 0 and os.environ.get("DCTR_GLUE_KERNELS", "1") != "0":
                # g_w = g^T feat as fixed-order column sums (csrc/head.hip k_colsum_part) instead of a [1, B] x [B, fm] GEMM
                g_wh = torch.empty((fm,), dtype=torch.float32, device=dev)
                ws = torch.empty((max(1, lib.dctr_relu_bwd_bias_workspace_floats(B, fm)),), dtype=torch.float32, device=dev)
                L.check(lib.dctr_rows_tdot(_ptr(feat), feat.stride(0), _ptr(g), B, fm, _ptr(g_wh), _ptr(ws), st),
                        "dctr_rows_tdot")
                g_wh = g_wh.reshape(w_head.shape)
            else:
                g_wh = torch.mm(g.reshape(1, B), feat).reshape(w_head.shape)
            wh = w_head.detach().contiguous().reshape(-1)
        rets = [None] * (2 * n)
        g_hidden = None
        first_gh = None
        for i in range(n - 1, -1, -1):
            O, nh, pf, o0 = ctx.geo[i]
            A = As[i]
            H = X0 if i == 0 else As[i - 1][:, :ctx.geo[i - 1][1]]
            Hr, ldh = (X0r, ldx) if i == 0 else _rows3(H, "CIN hidden input")
            h = Hr.shape[1]
            gA = torch.empty((B, O, D), dtype=torch.float32, device=dev)
            if ctx.has_head:
                gp, ld_gp, whp = _ptr(g), 1, ctypes.c_void_p(wh.data_ptr() + 4 * o0)
            else:
                gp, ld_gp, whp = ctypes.c_void_p(g.data_ptr() + 4 * o0), fm, None
            if B > 0:
                # (the relu's backward is applied while gA is assembled: the layer kernels then need no mask loads)
                L.check(lib.dctr_cin_pool_bwd(_ptr(g_hidden) if nh > 0 else None, gp, ld_gp, whp,
                                              _ptr(A) if ctx.relu else None, B, O, D, nh, pf, _ptr(gA), st),
                        "dctr_cin_pool_bwd")
            W2d = Ws[i].detach().reshape(O, -1)
            W2d = W2d if W2d.is_contiguous() else W2d.contiguous()
            # the hidden rows' gradient: the previous layer's g_hidden; layer 1's hidden state IS the field matrix -- its
            # two gradients meet in gx (the symmetric kernel writes them from different threads: a scratch + one add)
            gH = torch.empty((B, h, D), dtype=torch.float32, device=dev)
            gW = torch.empty((O, W2d.shape[1]), dtype=torch.float32, device=dev)
            gb = torch.empty((O,), dtype=torch.float32, device=dev) if ctx.has_bias[i] else None
            ws = torch.empty((max(1, lib.dctr_cin_bwd_workspace_floats(B, h, F, D, O)),), dtype=torch.float32, device=dev)
            L.check(lib.dctr_cin_layer_bwd(_ptr(gA), _ptr(A), O * D, 0, _ptr(Hr), ldh, _ptr(X0r), ldx, _ptr(W2d),
                                           B, h, F, D, O, _ptr(gH), h * D, _ptr(gx), ld_gx, int(i != n - 1), _ptr(gW),
                                           _ptr(gb), _ptr(ws), st), "dctr_cin_layer_bwd")
            rets[2 * i] = gW.reshape(ctx.w_shapes[i])
            rets[2 * i + 1] = gb
            if i == 0:
                first_gh = gH
            else:
                g_hidden = gH
        if B > 0:
            if ctx.flat and ctx_join:
                # gx[:, :F D] += first_gh and gx[:, F D:] = 0 in one launch (in place: every lane reads and writes its own words)
                fg = first_gh.reshape(B, F * D)
                L.check(lib.dctr_rows_join(_ptr(gx), ld_gx, _ptr(fg), fg.stride(0), F * D, None, 0, 0, _ptr(gx), ld_gx, B, st),
                        "dctr_rows_join")
            elif ctx.flat:
                gx[:, :F * D].add_(first_gh.reshape(B, F * D))
            else:
                gx.add_(first_gh)
        return (gx, None, None, None, None, g_wh) + tuple(rets)


class SENETFunction(torch.autograd.Function):
    """V = E * relu(W2 relu(W1 mean_d(E)))  (interaction.py:93-101)."""

    @staticmethod
    def forward(ctx, E, W1, W2):
        lib = L.lib()
        E, lde = _rows3(E, "SENET input")
        B, F, D = E.shape
        R = W1.shape[0]
        W1, W2 = W1.contiguous(), W2.contiguous()
        V = torch.empty((B, F, D), dtype=torch.float32, device=E.device)
        a = torch.empty((B, F), dtype=torch.float32, device=E.device)
        a1 = torch.empty((B, R), dtype=torch.float32, device=E.device)
        L.check(lib.dctr_senet_fwd(_ptr(E), lde, B, F, D, _ptr(W1), _ptr(W2), R, _ptr(V), _ptr(a), _ptr(a1),
                                   L.stream_handle(E.device)), "dctr_senet_fwd")
        ctx.save_for_backward(E, W1, W2, a, a1)
        return V

    @staticmethod
    def backward(ctx, gV):
        lib = L.lib()
        E, W1, W2, a, a1 = ctx.saved_tensors
        E, lde = _rows3(E, "SENET input")
        B, F, D = E.shape
        R = W1.shape[0]
        gV = gV.contiguous().float()
        gE = torch.empty((B, F, D), dtype=torch.float32, device=E.device)
        gW1, gW2 = torch.empty_like(W1), torch.empty_like(W2)
        ws = torch.empty((max(1, lib.dctr_senet_bwd_workspace_floats(B, F, R)),), dtype=torch.float32, device=E.device)
        L.check(lib.dctr_senet_bwd(_ptr(gV), _ptr(E), lde, B, F, D, _ptr(W1), _ptr(W2), R, _ptr(a), _ptr(a1), _ptr(gE),
                                   _ptr(gW1), _ptr(gW2), _ptr(ws), L.stream_handle(E.device)), "dctr_senet_bwd")
        return gE, gW1, gW2


def tournament_schedule(F, bilinear_type):
    """Pairs (i < j) of F fields in round-robin-tournament order: every round is a perfect matching, so workers
    handling different slots of a round never share a field.  Returns (rows [n, 4] = {i, j, w, k}, slots per round,
    pair_w [P], n_w).  k is the reference's pair index (itertools.combinations order)."""
    n = F + (F & 1)
    ring = list(range(n))
    rows = []
    for _ in range(n - 1):
        for s in range(n // 2):
            a, b = ring[s], ring[n - 1 - s]
            i, j = min(a, b), max(a, b)
            if j >= F:       # the dummy of an odd field count: idle slot
                rows.append((-1, -1, 0, 0))
                continue
            k = i * F - i * (i + 1) // 2 + (j - i - 1)
            w = 0 if bilinear_type == "all" else (i if bilinear_type == "each" else k)
            rows.append((i, j, w, k))
        ring = [ring[0]] + [ring[-1]] + ring[1:-1]
    P = F * (F - 1) // 2
    pair_w = [0] * P
    for (i, j, w, k) in rows:
        if i >= 0:
            pair_w[k] = w
    n_w = 1 if bilinear_type == "all" else (F if bilinear_type == "each" else P)
    return rows, n // 2, pair_w, n_w


def disjoint_groups(rows, width=8):
    """The pairs of a tournament schedule re-dealt in groups of ``width`` field-disjoint pairs (one per wave of a
    workgroup, a barrier per group): ``[n_groups][width][4]`` rows ``{i, j, w, k}``, ``i = -1`` for an idle entry.
    Greedy over the tournament order -- a round is a perfect matching, so only groups that straddle two rounds have to
    look ahead; deterministic."""
    rest = [r for r in rows if r[0] >= 0]
    groups = []
    while rest:
        used, grp, keep = set(), [], []
        for r in rest:
            if len(grp) < width and r[0] not in used and r[1] not in used:
                grp.append(r)
                used.update((r[0], r[1]))
            else:
                keep.append(r)
        rest = keep
        groups.append(grp + [(-1, -1, 0, 0)] * (width - len(grp)))
    return groups


def slab_ld(width):
    """Row stride (floats) of a wide [B, width] slab whose rows should start on 128-byte lines (DCTR_SLAB_ALIGN=0: dense)."""
    if os.environ.get("DCTR_SLAB_ALIGN", "1") == "0" or width < 1024:
        return int(width)
    return (int(width) + 31) // 32 * 32


class BilinearFunction(torch.autograd.Function):
    """(x_i W^T) * x_j for every pair, on E and optionally on a second input V with the same weights; the result is
    written in the DNN-input layout ``[V pairs | E pairs | dense]`` (fibinet.py:82-87)."""

    @staticmethod
    def forward(ctx, meta, E, V, dense, *weights):
        lib = L.lib()
        E, lde = _rows3(E, "Bilinear input")
        B, F, D = E.shape
        if D > 16:
            raise NotImplementedError("the gfx950 bilinear kernels support embedding_dim <= 16 (got %d)" % D)
        ldv = 0
        if V is not None:
            V, ldv = _rows3(V, "Bilinear second input")
        Wf = meta.flat_weights(weights)
        P = F * (F - 1) // 2
        npass = 2 if V is not None else 1
        n_dense = dense.shape[1] if dense is not None else 0
        if dense is not None and (dense.stride(1) != 1 or dense.dtype != torch.float32):
            dense = dense.float().contiguous()
        width = npass * P * D + n_dense
        # rows of the product slab start on a 128-byte line (round 6): a pair's D floats per sample are one 64-byte piece, and
        # with rows of 10 413 floats (41 652 bytes) every piece straddled two lines -- twice the memory requests in this
        # kernel's stores and in the backward kernels' reads of the gradient slab (mlp.WideLinearFunction returns it with the
        # same row stride).  The GEMMs behind take the view with its leading dimension.
        ld_out = slab_ld(width)
        out = torch.empty((B, ld_out), dtype=torch.float32, device=E.device)[:, :width]
        sched = meta.device_tables(E.device)
        L.check(lib.dctr_bilinear_fwd(_ptr(E), lde, _ptr(V), ldv, _ptr(Wf), _ptr(sched[2]), sched[2].shape[0], P, F, D, B,
                                      _ptr(out), ld_out, _ptr(dense), dense.stride(0) if dense is not None else 0,
                                      n_dense, npass * P * D, L.stream_handle(E.device)), "dctr_bilinear_fwd")
        ctx.meta, ctx.n_w_in = meta, len(weights)
        ctx.has_v, ctx.has_dense = V is not None, dense is not None
        ctx.save_for_backward(E, V, Wf)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        meta = ctx.meta
        E, V, Wf = ctx.saved_tensors
        E, lde = _rows3(E, "Bilinear input")
        B, F, D = E.shape
        ldv = 0
        if V is not None:
            V, ldv = _rows3(V, "Bilinear second input")
        P = F * (F - 1) // 2
        npass = 2 if V is not None else 1
        gout = gout.float()
        if gout.stride(1) != 1:
            gout = gout.contiguous()
        dev = E.device
        gE = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        gV = torch.empty((B, F, D), dtype=torch.float32, device=dev) if V is not None else None
        gW = torch.empty((meta.n_w, D, D), dtype=torch.float32, device=dev)
        ws = torch.empty((max(1, lib.dctr_bilinear_bwd_workspace_floats(B, P, D)),), dtype=torch.float32, device=dev)
        sched = meta.device_tables(dev)
        L.check(lib.dctr_bilinear_bwd(_ptr(E), lde, _ptr(V), ldv, _ptr(Wf), _ptr(sched[0]), meta.n_sched, meta.slots,
                                      _ptr(sched[1]), meta.n_w, P, F, D, B, _ptr(gout), gout.stride(0), _ptr(gE), _ptr(gV),
                                      _ptr(gW), _ptr(ws), _ptr(sched[2]), sched[2].shape[0], L.stream_handle(dev)),
                "dctr_bilinear_bwd")
        g_dense = gout[:, npass * P * D:] if ctx.has_dense else None
        return (None, gE, gV, g_dense) + tuple(gW[i] for i in range(ctx.n_w_in))


class BilinearStackedFunction(torch.autograd.Function):
    """``(x_i W_k^T) * x_j`` for every pair k with the weights given as ONE ``[n_w, D, D]`` tensor (no per-weight
    parameters to re-seat): the bilinear kernels of csrc/pairwise.hip behind OutterProductLayer's 'mat' kernel."""

    @staticmethod
    def forward(ctx, meta, E, Wf):
        lib = L.lib()
        E, lde = _rows3(E, "pairwise input")
        B, F, D = E.shape
        if D > 16:
            raise NotImplementedError("the gfx950 bilinear kernels support embedding_dim <= 16 (got %d)" % D)
        Wf = Wf.detach().float().contiguous()
        P = F * (F - 1) // 2
        out = torch.empty((B, P * D), dtype=torch.float32, device=E.device)
        sched = meta.device_tables(E.device)
        L.check(lib.dctr_bilinear_fwd(_ptr(E), lde, None, 0, _ptr(Wf), _ptr(sched[2]), sched[2].shape[0], P, F, D, B,
                                      _ptr(out), P * D, None, 0, 0, P * D, L.stream_handle(E.device)),
                "dctr_bilinear_fwd")
        ctx.meta = meta
        ctx.save_for_backward(E, Wf)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        meta = ctx.meta
        E, Wf = ctx.saved_tensors
        E, lde = _rows3(E, "pairwise input")
        B, F, D = E.shape
        P = F * (F - 1) // 2
        gout = gout.float()
        if gout.stride(1) != 1:
            gout = gout.contiguous()
        dev = E.device
        gE = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        gW = torch.empty((meta.n_w, D, D), dtype=torch.float32, device=dev)
        ws = torch.empty((max(1, lib.dctr_bilinear_bwd_workspace_floats(B, P, D)),), dtype=torch.float32, device=dev)
        sched = meta.device_tables(dev)
        L.check(lib.dctr_bilinear_bwd(_ptr(E), lde, None, 0, _ptr(Wf), _ptr(sched[0]), meta.n_sched, meta.slots,
                                      _ptr(sched[1]), meta.n_w, P, F, D, B, _ptr(gout), gout.stride(0), _ptr(gE), None,
                                      _ptr(gW), _ptr(ws), _ptr(sched[2]), sched[2].shape[0], L.stream_handle(dev)),
                "dctr_bilinear_bwd")
        return None, gE, gW


class BilinearMeta(object):
    """Host-side tables of a BilinearInteraction layer: the tournament schedule and the flat weight slab."""

    def __init__(self, F, bilinear_type):
        rows, self.slots, pair_w, self.n_w = tournament_schedule(F, bilinear_type)
        self.n_sched = len(rows)
        self._rows, self._pair_w = rows, pair_w
        self._dev = None
        self._slab = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_dev"] = None
        d["_slab"] = None
        d["_wide"] = None
        return d

    def device_tables(self, device):
        if self._dev is None or self._dev[0].device != torch.device(device):
            # [2]: the forward's order -- by output position k: the four waves of a workgroup then write neighbouring
            # 64-byte pieces of a sample's row at about the same time, and L2 evicts whole lines (the tournament order
            # scattered them: measured 126 us for the 170 MB of FiBiNET's DNN input at the Criteo shape)
            by_k = sorted((r for r in self._rows if r[0] >= 0), key=lambda r: r[3])
            self._dev = (torch.tensor(self._rows, dtype=torch.int32, device=device).reshape(-1, 4).contiguous(),
                         torch.tensor(self._pair_w, dtype=torch.int32, device=device),
                         torch.tensor(by_k, dtype=torch.int32, device=device).reshape(-1, 4).contiguous())
        return self._dev

    def wide_tables(self, device):
        """(groups ``[n_groups, 8, 4]`` int32, pair_w) on ``device`` for dctr_bilinear_wide_bwd."""
        dev = getattr(self, "_wide", None)
        if dev is None or dev[0].device != torch.device(device):
            groups = disjoint_groups(self._rows)
            self._wide = (torch.tensor(groups, dtype=torch.int32, device=device).reshape(-1, 8, 4).contiguous(),
                          self.device_tables(device)[1])
        return self._wide

    def flat_weights(self, weights):
        """``[n_w, D, D]`` slab holding the layer's nn.Linear weights.  The parameters are re-seated ONCE as slices of
        one contiguous slab (values preserved), after which this is a pointer check; the slab pointer stays stable
        (hipGraph-safe) until someone re-allocates the parameters (``.to()``), which is detected here."""
        slab = self._slab
        D = weights[0].shape[0]
        step = D * D * 4
        if slab is not None and slab.shape[0] == len(weights) and slab.device == weights[0].device and \
                all(w.data_ptr() == slab.data_ptr() + i * step for i, w in enumerate(weights)):
            return slab
        slab = torch.stack([w.detach() for w in weights]).contiguous()
        for i, w in enumerate(weights):
            w.data = slab[i]
        self._slab = slab
        return slab


class InnerProductFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, reduce_sum):
        lib = L.lib()
        E, lde = _rows3(E, "InnerProduct input")
        B, F, D = E.shape
        P = F * (F - 1) // 2
        per = 1 if reduce_sum else D
        out = torch.empty((B, P, per), dtype=torch.float32, device=E.device)
        if P > 0:
            L.check(lib.dctr_inner_product_fwd(_ptr(E), lde, B, F, D, int(bool(reduce_sum)), _ptr(out), P * per,
                                               L.stream_handle(E.device)), "dctr_inner_product_fwd")
        ctx.reduce_sum = bool(reduce_sum)
        ctx.save_for_backward(E)
        return out

    @staticmethod
    def backward(ctx, gp):
        lib = L.lib()
        (E,) = ctx.saved_tensors
        E, lde = _rows3(E, "InnerProduct input")
        B, F, D = E.shape
        P = F * (F - 1) // 2
        per = 1 if ctx.reduce_sum else D
        gp = gp.contiguous().float()
        gE = torch.zeros((B, F, D), dtype=torch.float32, device=E.device)
        if P > 0:
            L.check(lib.dctr_inner_product_bwd(_ptr(E), lde, B, F, D, int(ctx.reduce_sum), _ptr(gp), P * per, _ptr(gE),
                                               F * D, L.stream_handle(E.device)), "dctr_inner_product_bwd")
        return gE, None


# ---- CrossNet, vector parameterisation (csrc/cross.hip) -----------------------------------------------------
class CrossNetVecFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, kernels, bias):
        lib = L.lib()
        X = _rows_f32(X, "CrossNet input")
        B, W = X.shape
        Lyr = kernels.shape[0]
        if W > 2048:
            raise NotImplementedError("the gfx950 CrossNet kernel supports in_features <= 2048 (got %d)" % W)
        k2, b2 = kernels.reshape(Lyr, W).contiguous(), bias.reshape(Lyr, W).contiguous()
        Y = torch.empty((B, W), dtype=torch.float32, device=X.device)
        L.check(lib.dctr_crossnet_vec_fwd(_ptr(X), X.stride(0) if B > 1 else W, B, W, Lyr, _ptr(k2), _ptr(b2), _ptr(Y), W,
                                          L.stream_handle(X.device)), "dctr_crossnet_vec_fwd")
        ctx.save_for_backward(X, k2, b2)
        ctx.kshape, ctx.bshape = kernels.shape, bias.shape
        return Y

    @staticmethod
    def backward(ctx, gY):
        lib = L.lib()
        X, k2, b2 = ctx.saved_tensors
        B, W = X.shape
        Lyr = k2.shape[0]
        gY = _rows_f32(gY, "CrossNet output gradient")
        gX = torch.empty((B, W), dtype=torch.float32, device=X.device)
        gk, gb = torch.empty_like(k2), torch.empty_like(b2)
        ws = torch.empty((max(1, lib.dctr_crossnet_vec_bwd_workspace_floats(B, W, Lyr)),), dtype=torch.float32,
                         device=X.device)
        L.check(lib.dctr_crossnet_vec_bwd(_ptr(X), X.stride(0) if B > 1 else W, B, W, Lyr, _ptr(k2), _ptr(b2), _ptr(gY),
                                          gY.stride(0) if B > 1 else W, _ptr(gX), W, _ptr(gk), _ptr(gb), _ptr(ws),
                                          L.stream_handle(X.device)), "dctr_crossnet_vec_bwd")
        return gX, gk.reshape(ctx.kshape), gb.reshape(ctx.bshape)


class CrossNetMatFunction(torch.autograd.Function):
    """CrossNet, matrix parameterisation (reference interaction.py:448-451): ``x_{l+1} = x_0 * (x_l W_l^T + b_l) + x_l``
    for all layers in ONE forward launch (16 samples per workgroup, x_0 / x_l in LDS, fp32 MFMA) and, for the backward,
    one data launch + the tower's weight-gradient kernels (csrc/mlp.hip: dctr_crossnet_mat_fwd / _bwd)."""

    @staticmethod
    def _desc(Wp, ld_w, bias, hs, us, gWs, gbs, W):
        desc = L.Mlp()
        desc.n_layers = len(Wp)
        for l in range(len(Wp)):
            e = desc.layer[l]
            e.W, e.bias = Wp[l].data_ptr(), bias[l].data_ptr()
            e.h, e.dh = hs[l].data_ptr(), us[l].data_ptr()
            e.gW = gWs[l].data_ptr() if gWs is not None else None
            e.gbias = gbs[l].data_ptr() if gbs is not None else None
            e.K = e.N = W
            e.ld_w, e.ld_h, e.relu = ld_w, hs[l].stride(0), 0
        desc.w_out = desc.g_w_out = None
        return desc

    @staticmethod
    def forward(ctx, X, kernels, bias):
        lib = L.lib()
        L.require_gpu(X, "CrossNet input")
        B, W = X.shape
        Lyr = kernels.shape[0]
        ld = (W + 3) // 4 * 4
        if X.dtype != torch.float32 or X.stride(1) != 1 or X.stride(0) % 4 or X.data_ptr() % 16 or X.stride(0) < W:
            buf = torch.zeros((B, ld), dtype=torch.float32, device=X.device)
            buf[:, :W].copy_(X)
            X = buf
        # weights with rows padded to a multiple of 4 floats (zeros), one [L, W, ld] block
        Wpad = torch.zeros((Lyr, W, ld), dtype=torch.float32, device=X.device)
        Wpad[:, :, :W].copy_(kernels.detach())
        b2 = bias.detach().reshape(Lyr, W).contiguous()
        hs = [torch.empty((B, ld), dtype=torch.float32, device=X.device) for _ in range(Lyr)]
        us = [torch.empty((B, ld), dtype=torch.float32, device=X.device) for _ in range(Lyr)]
        desc = CrossNetMatFunction._desc([Wpad[l] for l in range(Lyr)], ld, [b2[l] for l in range(Lyr)], hs, us, None,
                                         None, W)
        L.check(lib.dctr_crossnet_mat_fwd(ctypes.byref(desc), _ptr(X), X.stride(0), B, L.stream_handle(X.device)),
                "dctr_crossnet_mat_fwd")
        ctx.save_for_backward(X, Wpad, b2, *(hs + us))
        ctx.dims = (B, W, Lyr, ld)
        ctx.kshape, ctx.bshape = kernels.shape, bias.shape
        return hs[-1][:, :W]

    @staticmethod
    def backward(ctx, gY):
        lib = L.lib()
        B, W, Lyr, ld = ctx.dims
        saved = ctx.saved_tensors
        X, Wpad, b2 = saved[0], saved[1], saved[2]
        hs, us = list(saved[3:3 + Lyr]), list(saved[3 + Lyr:3 + 2 * Lyr])
        g = torch.zeros((B, ld), dtype=torch.float32, device=X.device)
        g[:, :W].copy_(gY)
        gW = torch.empty((Lyr, W, ld), dtype=torch.float32, device=X.device)
        gb = torch.empty((Lyr, W), dtype=torch.float32, device=X.device)
        gX = torch.empty((B, ld), dtype=torch.float32, device=X.device)
        desc = CrossNetMatFunction._desc([Wpad[l] for l in range(Lyr)], ld, [b2[l] for l in range(Lyr)], hs, us,
                                         [gW[l] for l in range(Lyr)], [gb[l] for l in range(Lyr)], W)
        ws = torch.empty((max(1, lib.dctr_crossnet_mat_bwd_workspace_floats(ctypes.byref(desc), B)),),
                         dtype=torch.float32, device=X.device)
        L.check(lib.dctr_crossnet_mat_bwd(ctypes.byref(desc), _ptr(X), X.stride(0), B, _ptr(g), ld, _ptr(gX), ld, _ptr(ws),
                                          L.stream_handle(X.device)), "dctr_crossnet_mat_bwd")
        return gX[:, :W], gW[:, :, :W].reshape(ctx.kshape), gb.reshape(ctx.bshape)


class CrossNetMixFunction(torch.autograd.Function):
    """CrossNetMix of DCN-Mix (reference interaction.py:499-534): per cross layer a mixture of low-rank experts,
    ``x_{l+1} = x_0 * (sum_e softmax(x_l G^T)_e * tanh(tanh(x_l V_e) C_e^T) U_e^T + b) + x_l``.  All layers in ONE
    forward launch (csrc/mlp.hip ``dctr_crossnet_mix_fwd``: three dense fp32-MFMA layers per cross layer on a 16-sample
    tile kept in LDS) and one backward-data launch + the tower's weight-gradient kernels.  The weights travel packed:
    ``W1 = [V (E*R rows) | G (E rows)] x W``, ``W2 = blockdiag(C_e)``, ``W3[w, e*R + r] = U_e[w, r]``."""

    @staticmethod
    def _r4(n):
        return (int(n) + 3) // 4 * 4

    @staticmethod
    def _pack(U, V, C, G, bias):
        Lc, E, W, R = U.shape
        ER = E * R
        ldW, ldE = CrossNetMixFunction._r4(W), CrossNetMixFunction._r4(ER)
        dev = U.device
        W1 = torch.zeros((Lc, ER + E, ldW), dtype=torch.float32, device=dev)
        W1[:, :ER, :W] = V.permute(0, 1, 3, 2).reshape(Lc, ER, W)       # row e*R + r = V_e[:, r]
        W1[:, ER:, :W] = G.unsqueeze(0)
        W2 = torch.zeros((Lc, ER, ldE), dtype=torch.float32, device=dev)
        for e in range(E):
            W2[:, e * R:(e + 1) * R, e * R:(e + 1) * R] = C[:, e]         # v2[e, r] = sum_s C[e, r, s] v1[e, s]
        W3 = torch.zeros((Lc, W, ldE), dtype=torch.float32, device=dev)
        W3[:, :, :ER] = U.permute(0, 2, 1, 3).reshape(Lc, W, ER)        # W3[w, e*R + r] = U_e[w, r]
        return W1, W2, W3, bias.reshape(Lc, W).contiguous()

    @staticmethod
    def _desc(W1, W2, W3, b2, bufs, grads, dims):
        B, W, Lc, E, R = dims
        ER = E * R
        desc = L.Mlp()
        desc.n_layers = 3 * Lc
        for lc in range(Lc):
            for k, (Wt, K, N) in enumerate(((W1[lc], W, ER + E), (W2[lc], ER, ER), (W3[lc], ER, W))):
                e = desc.layer[3 * lc + k]
                h, dh = bufs[3 * lc + k]
                e.W, e.bias = Wt.data_ptr(), (b2[lc].data_ptr() if k == 2 else None)
                e.h, e.dh = h.data_ptr(), dh.data_ptr()
                e.K, e.N, e.ld_w, e.ld_h, e.relu = K, N, Wt.stride(0), h.stride(0), 0
                if grads is not None:
                    gW, gb = grads[3 * lc + k]
                    e.gW, e.gbias = gW.data_ptr(), (gb.data_ptr() if gb is not None else None)
                else:
                    e.gW = e.gbias = None
        desc.w_out = desc.g_w_out = None
        return desc

    @staticmethod
    def forward(ctx, X, U, V, C, G, bias):
        lib = L.lib()
        L.require_gpu(X, "CrossNetMix input")
        B, W = X.shape
        Lc, E, _, R = U.shape
        ER, r4 = E * R, CrossNetMixFunction._r4
        ld = r4(W)
        if X.dtype != torch.float32 or X.stride(1) != 1 or X.stride(0) % 4 or X.data_ptr() % 16 or X.stride(0) < W:
            buf = torch.zeros((B, ld), dtype=torch.float32, device=X.device)
            buf[:, :W].copy_(X)
            X = buf
        W1, W2, W3, b2 = CrossNetMixFunction._pack(U.detach(), V.detach(), C.detach(), G.detach(), bias.detach())
        bufs = []
        for lc in range(Lc):
            for n in (ER + E, ER, W):
                bufs.append((torch.empty((B, r4(n)), dtype=torch.float32, device=X.device),
                             torch.empty((B, r4(n)), dtype=torch.float32, device=X.device)))
        dims = (B, W, Lc, E, R)
        desc = CrossNetMixFunction._desc(W1, W2, W3, b2, bufs, None, dims)
        L.check(lib.dctr_crossnet_mix_fwd(ctypes.byref(desc), E, R, _ptr(X), X.stride(0), B, L.stream_handle(X.device)),
                "dctr_crossnet_mix_fwd")
        ctx.save_for_backward(X, W1, W2, W3, b2, *[t for pair in bufs for t in pair])
        ctx.dims = dims
        ctx.shapes = (U.shape, V.shape, C.shape, G.shape, bias.shape)
        return bufs[-1][0][:, :W]

    @staticmethod
    def backward(ctx, gY):
        lib = L.lib()
        B, W, Lc, E, R = ctx.dims
        ER, r4 = E * R, CrossNetMixFunction._r4
        saved = ctx.saved_tensors
        X, W1, W2, W3, b2 = saved[:5]
        flat = saved[5:]
        bufs = [(flat[2 * i], flat[2 * i + 1]) for i in range(3 * Lc)]
        dev = X.device
        ld = r4(W)
        g = torch.zeros((B, ld), dtype=torch.float32, device=dev)
        g[:, :W].copy_(gY)
        gW1, gW2, gW3 = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(W3)
        gb = torch.empty((Lc, W), dtype=torch.float32, device=dev)
        grads = []
        for lc in range(Lc):
            grads += [(gW1[lc], None), (gW2[lc], None), (gW3[lc], gb[lc])]
        gX = torch.empty((B, ld), dtype=torch.float32, device=dev)
        desc = CrossNetMixFunction._desc(W1, W2, W3, b2, bufs, grads, ctx.dims)
        ws = torch.empty((max(1, lib.dctr_crossnet_mix_bwd_workspace_floats(ctypes.byref(desc), B)),),
                         dtype=torch.float32, device=dev)
        L.check(lib.dctr_crossnet_mix_bwd(ctypes.byref(desc), E, R, _ptr(X), X.stride(0), B, _ptr(g), ld, _ptr(gX), ld,
                                          _ptr(ws), L.stream_handle(dev)), "dctr_crossnet_mix_bwd")
        # unpack: V / G from the rows of gW1, the diagonal blocks of gW2, U from gW3
        gV = gW1[:, :ER, :W].reshape(Lc, E, R, W).permute(0, 1, 3, 2)
        gG = gW1[:, ER:, :W].sum(0)
        gC = torch.stack([gW2[:, e * R:(e + 1) * R, e * R:(e + 1) * R] for e in range(E)], dim=1)
        gU = gW3[:, :, :ER].reshape(Lc, W, E, R).permute(0, 2, 1, 3)
        sU, sV, sC, sG, sb = ctx.shapes
        return gX[:, :W], gU.reshape(sU), gV.reshape(sV), gC.reshape(sC), gG.reshape(sG), gb.reshape(sb)
