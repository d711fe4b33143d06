"""The train step of a DeepFM / WDL-shaped model as five device launches on two queues, enqueued directly through
the C ABI (round 4).

Reference: one ``fit()`` inner iteration (basemodel.py:242-262) of a model whose logit is
``linear(X) [+ FM] + tower(combined_dnn_input) + bias`` (deepfm.py:67-86, wdl.py) -- 52 ``aten::embedding`` calls, the
FM / concat chain, 9 GEMMs, ~30 elementwise launches forward, the same again backward, then ``optim.step()`` over every
table.  Here:

    side queue :  dctr_embed_ids + dctr_embed_segments (id-only pre-pass of the update; needs X alone)
    main queue :  dctr_embed_tower_train_step   gather + linear + FM + tower + head + BCE + backward-data, ONE launch:
                                                every workgroup gathers the rows of its own 16 samples (csrc/mlp.hip)
    main queue :  dctr_mlp_train_wgrad          weight gradients + fixed-order reduction + the dense optimizer step
    side queue :  dctr_embed_update             behind the first launch: sort-based row update, optimizer inside
    (main waits for side: the next step's gather reads the rows this update writes)

Rounds 1-3 assembled the same step out of ``torch.autograd.Function``s (ops.EmbedFunction, mlp.TowerHeadFunction) with a
gather kernel of its own in front of the tower: gather -> tower -> [queue hop] -> update -> [queue hop] -> next gather was
the step's critical cycle (12.6 + 43 + 12 + 21 + 12 us of 99).  With the gather inside the tower launch that kernel, its
output's write-then-read and one hop are gone from the cycle.  All buffers of a batch size are allocated once and reused
by every step (every consumer of a step's buffers has finished before the next tower launch starts -- it waits for both
queues), so a captured multi-step hipGraph touches the same 40 MB of activations in every step instead of 40 MB per
captured step.

The arithmetic is the two-launch path's, in its order: parameters after any number of steps are bit-identical
(tests/test_gpu_step_engine.py).

Round 6, an OPT-IN topology (``DCTR_STEP_TOPOLOGY=weights_flag``; bit-identical, measured slower than the default:
``GatherStep.topology``):

    main queue :  tower(k) -> dctr_embed_update(k) -> tower(k+1) ...
    side queue :  [after tower(k)] dctr_mlp_train_wgrad_sync(k) -> [after update(k)] pre-pass of batch k+1

What tower(k+1) needs from the side queue -- the dense parameters stepped by the weight-gradient launch of step k -- it waits
for INSIDE the kernel, after it has staged its X tile and gathered its table rows: a word in memory the weight-gradient launch
advances when its last reducer has stored the parameters (include/dctr.h, DCTR_SYNC_W_GEN).  The reduction is folded into the
weight-gradient launch (a tile's last arriver sums the partial slabs in slab order: the same bits).  tower(k+1) then runs
beside the weight gradients of step k, which read step k's activations: the activation buffers exist twice and alternate.
"""
import contextlib
import ctypes
import os

import torch

from . import lib as L
from . import mlp as _mlp
from . import streams as _streams


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _r4(n):
    return (int(n) + 3) // 4 * 4


class _Buffers(object):
    """Everything one batch size needs, allocated once: activations, gradients, workspaces, the tower descriptor.
    ``alt``: a second set of what the tower launch writes and the weight-gradient launch reads (out, gx, fm_s, g_logit,
    hs, dhs, ws, desc) -- the weights_flag topology alternates between the two; everything else is shared."""
    __slots__ = ("B", "out", "gx", "fm_s", "g_logit", "hs", "dhs", "ws", "ids_t", "parts_t", "desc", "upd_ws", "upd_n",
                 "keep", "pinned", "den_t", "amax", "alt", "cnt")


class GatherStep(object):
    """See the module docstring.  Built by ``BaseModel._fused_step_state`` for models that declare
    ``_gather_step = True`` (their ``logit_parts`` is ``[linear, (fm), tower]`` over one fused lookup)."""

    def __init__(self, model, slab):
        self.model, self.slab = model, slab
        self._bufs = {}
        spec = _mlp.tower_layers(model.dnn, model.dnn_linear)
        self.layers, self.w_out = spec
        self.want_fm = bool(getattr(model, "use_fm", False)) and len(model.model_plan().deep) > 0
        self._prepassed = None    # token of the batch whose pre-pass the previous step already enqueued
        self._events, self._ev_i = None, 0
        self.timing = None        # (start, end) events around the update on its queue: bench.py's in-step roofline
        self.timing_tower = None  # (start, end) events around the gather + tower launch on the main queue
        self.stamps = None        # int64 [n, 4 | 5] device tensor: bench.py's graph-replayed kernel durations -- step k dates the
        self.stamp_i = 0          # boundaries around its tower launch (columns 0, 1) and its update (2, 3): dctr_stamp;
                                  # column 4: one more stamp right in front of column 0's (the cost of a stamp launch)
        self.steps_run = 0        # steps this engine enqueued (eager or captured): who asks whether it really ran
        self._sync = None         # the weights_flag topology's sync block (include/dctr.h DCTR_SYNC_W_GEN / T_GEN)
        self._parity = 0          # which activation set the next step writes
        self._pool, self._pool_i = None, 0

    # ---- applicability ------------------------------------------------------------------------------------------
    @staticmethod
    def enabled():
        return os.environ.get("DCTR_STEP_ENGINE", "1") != "0"

    def supports(self, xb, yb):
        """True when this batch can take the engine (else the caller runs the autograd-assembled fused step)."""
        plan = self.model.model_plan()
        if xb.dim() != 2 or xb.dtype != torch.float32 or xb.stride(1) != 1 or xb.shape[1] < plan.n_xcols:
            return False
        if yb.numel() != xb.shape[0]:
            return False
        if plan.update[0] not in ("sgd", "adagrad") or not plan.segments_enabled() or self.slab.lam is not None:
            return False
        if getattr(plan, "sharder", None) is not None or getattr(plan, "exchange", None) is not None:
            return False
        b = self._buffers(xb.shape[0], xb.device)
        return b is not None

    # ---- buffers --------------------------------------------------------------------------------------------------
    def _buffers(self, B, dev):
        plan, slab = self.model.model_plan(), self.slab
        plan.bind(dev)
        key = (int(B), str(dev), plan.version)
        hit = self._bufs.get(key)
        if hit is not None:
            self._bufs[key] = self._bufs.pop(key)      # (most recently used last)
            return hit if hit is not False else None
        lib = L.lib()
        b = _Buffers()
        b.B = int(B)
        f32 = dict(dtype=torch.float32, device=dev)
        Ws, lds, bp = [], [], []
        for (W, bias, _) in self.layers:
            w, ld = _mlp._rows4(W)
            if w is not W:          # (not slab-seated: the autograd route copes with it)
                self._bufs[key] = False
                return None
            Ws.append(w)
            lds.append(ld)
            bp.append(bias)
        b.hs = [torch.empty((B, _r4(W.shape[0])), **f32) for W in Ws]
        b.dhs = [torch.empty_like(h) for h in b.hs]
        gWs = [slab.grad_of(W) for (W, _, _) in self.layers]
        gbs = [slab.grad_of(bias) if bias is not None else None for (_, bias, _) in self.layers]
        g_wo = slab.grad_of(self.w_out).reshape(-1)
        meta = _mlp._Meta([r for (_, _, r) in self.layers], True, plan.width)
        b.desc = L.Mlp()
        _mlp._fill(b.desc, meta, Ws, lds, bp, b.hs, b.dhs, gWs, gbs, self.w_out.reshape(-1), g_wo)
        ok = plan.unit_path and plan.update_kernel_ok(B) and \
            lib.dctr_embed_tower_train_supported(ctypes.byref(plan.cplan), ctypes.byref(b.desc), int(B)) == 1
        if not ok:
            self._bufs[key] = False
            return None
        b.out = torch.empty((B, plan.ld_out), **f32)
        b.gx = torch.empty((B, plan.ld_out), **f32)
        b.fm_s = torch.empty((B, _r4(plan.emb_dim)), **f32) if self.want_fm else None
        b.g_logit = torch.empty((B,), **f32)
        b.ws = torch.empty((max(1, lib.dctr_mlp_train_workspace_floats(ctypes.byref(b.desc), int(B))),), **f32)
        b.ids_t = torch.empty((plan.n_vcols, B), dtype=torch.int32, device=dev)
        b.parts_t = torch.empty((plan.n_vcols, B), dtype=torch.int16, device=dev)
        # (general units -- pooled fields, shared tables: mean pooling's divisors, written with the ids, and max pooling's
        # arg-max positions, written by the tower launch's gather stage)
        b.den_t, b.amax = plan.step_buffers(B, dev)
        b.upd_ws, b.upd_n = plan.update_workspace(B, dev, always=True)
        b.keep = (Ws, gWs, gbs, g_wo)
        b.pinned = False
        # the second activation set (weights_flag topology) and the weight-gradient launch's arrival counters
        a = _Buffers()
        a.B = b.B
        a.hs = [torch.empty_like(h) for h in b.hs]
        a.dhs = [torch.empty_like(h) for h in b.hs]
        a.desc = L.Mlp()
        _mlp._fill(a.desc, meta, Ws, lds, bp, a.hs, a.dhs, gWs, gbs, self.w_out.reshape(-1), g_wo)
        a.out, a.gx = torch.empty_like(b.out), torch.empty_like(b.gx)
        a.fm_s = torch.empty_like(b.fm_s) if b.fm_s is not None else None
        a.g_logit, a.ws = torch.empty_like(b.g_logit), torch.empty_like(b.ws)
        b.alt = a
        b.cnt = torch.zeros((max(1, lib.dctr_mlp_train_wgrad_counters(ctypes.byref(b.desc), int(B))),), dtype=torch.int32,
                            device=dev)
        # Least recently used goes first -- but never a set a hipGraph was captured on: the graph holds raw addresses of
        # out / gx / hs / ws / ids_t / the update workspace and nothing else keeps them alive (a fit() over shards of many
        # ragged tail sizes used to push the full-batch set out from under its still-replaying graph: round-4 advisor
        # finding).  Pinned sets of an older plan version are dead (their graphs re-capture) and may go.
        if len(self._bufs) >= 8:
            for k in list(self._bufs):
                v = self._bufs[k]
                if v is False or not v.pinned or k[2] != plan.version:
                    self._bufs.pop(k)
                    break
        self._bufs[key] = b
        return b

    # ---- one step -------------------------------------------------------------------------------------------------
    def _prepass(self, b, cplan, xb, B, stream_handle):
        """ids + partition tags from X, then every (unit, partition)'s entries found and sorted: all the update needs
        that is not a gradient."""
        lib = L.lib()
        plan = self.model.model_plan()
        units, n_units = plan.units_ptr(), plan.n_grid_units
        plan.point_step_buffers(b.den_t, b.amax)
        L.check(lib.dctr_embed_ids(cplan, units, n_units, _ptr(xb), xb.stride(0), B, _ptr(b.ids_t), _ptr(b.parts_t),
                                   stream_handle), "dctr_embed_ids")
        L.check(lib.dctr_embed_segments(cplan, units, n_units, plan.max_vocab, _ptr(b.ids_t), _ptr(b.parts_t), B,
                                        _ptr(b.upd_ws), b.upd_n, stream_handle), "dctr_embed_segments")

    def step(self, xb, yb, mode, next_xb=None):
        """Enqueue one train step on (xb, yb); returns (loss, y_pred) device tensors.  ``mode``: the dense optimizer's
        (kind, lr, eps).  ``next_xb``: the batch of the step that follows, when the caller knows it (the captured steps of
        a multi-step hipGraph): its pre-pass is enqueued on the side queue right behind this step's update, so that it
        runs in the shadow of the queue hop back to the main queue instead of beside the next tower launch (measured:
        the pre-pass's 1118 workgroups beside the tower cost the tower 4 of its 53 us, profiles/r04_*timeline*)."""
        if xb.device.type == "cuda" and self.topology() == "weights_flag":
            return self._step_flag(xb, yb, mode, next_xb)
        lib = L.lib()
        model, slab = self.model, self.slab
        plan = model.model_plan()
        dev = xb.device
        cuda = dev.type == "cuda"
        B = xb.shape[0]
        b = self._buffers(B, dev)
        self.steps_run += 1
        if cuda and torch.cuda.is_current_stream_capturing():
            b.pinned = True           # a graph now holds this set's addresses: exempt from eviction
        cplan = plan.bind(dev)
        y = yb.reshape(-1)
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.float().contiguous()
        y_pred = torch.empty((B,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        kind = plan.update[0]
        opt = L.UPD_ADAGRAD if kind == "adagrad" else L.UPD_SGD
        lr = float(plan.update[1])
        eps = float(plan.update[2]) if kind == "adagrad" else 0.0
        if not slab.begin_inline_step(mode[0], mode[1], mode[2] if len(mode) > 2 else 0.0):
            raise RuntimeError("the gather step needs a plain SGD / Adagrad dense optimizer")
        inline = slab.inline
        bias = model.out.bias
        g_bias = slab.grad_of(bias)
        lw = plan.wide_dense_weight
        g_wd = slab.grad_of(lw) if lw is not None else None
        err = plan.err_flag(dev)
        ws_u = b.upd_ws
        if getattr(ws_u, "_dctr_owner", None) is not None:      # an abandoned pre-pass of the autograd route left counts
            ws_u.zero_()
            ws_u._dctr_owner = None
            self._prepassed = None
        units, n_units = plan.units_ptr(), plan.n_grid_units
        plan.point_step_buffers(b.den_t, b.amax)
        ld = plan.ld_out
        ld_s = b.fm_s.stride(0) if b.fm_s is not None else 0
        serial = os.environ.get("DCTR_STEP_TOPOLOGY", "update_side") == "serial" or not cuda
        main = torch.cuda.current_stream(dev) if cuda else None
        side = _streams.side_stream(dev, "seg") if (cuda and not serial) else None
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        # (did the previous step already enqueue THIS batch's pre-pass?  Only ever inside one hipGraph capture: the token
        # names the batch's memory, the buffers and the capture)
        token = (xb.data_ptr(), xb.stride(0), B, id(b), bool(cuda and torch.cuda.is_current_stream_capturing()))
        have_prepass = self._prepassed is not None and self._prepassed == token and token[4]
        self._prepassed = None
        try:
            if not have_prepass:
                if side is not None:
                    side.wait_stream(main)   # X is complete; (first step of a capture: the side queue joins the capture)
                with on_side():
                    self._prepass(b, cplan, xb, B, L.stream_handle(dev))
            mh = L.stream_handle(dev)
            if self.timing_tower is not None:
                self.timing_tower[0].record(main)
            stamp = None
            if self.stamps is not None and cuda:
                row = self.stamp_i % self.stamps.shape[0]
                self.stamp_i += 1
                ncol = int(self.stamps.shape[1])
                stamp = [ctypes.c_void_p(self.stamps.data_ptr() + 8 * (ncol * row + c)) for c in range(ncol)]
                if ncol > 4:     # (two stamps back to back: what one stamp launch itself costs the queue)
                    L.check(lib.dctr_stamp(stamp[4], mh), "dctr_stamp")
                L.check(lib.dctr_stamp(stamp[0], mh), "dctr_stamp")
            L.check(lib.dctr_embed_tower_train_step(cplan, _ptr(xb), xb.stride(0), ctypes.byref(b.desc), B,
                                                    1 if self.want_fm else 0, _ptr(bias), _ptr(y), _ptr(y_pred),
                                                    _ptr(b.g_logit), _ptr(b.gx), ld, _ptr(b.out), ld, _ptr(b.fm_s), ld_s,
                                                    _ptr(err), _ptr(b.ws), mh), "dctr_embed_tower_train_step")
            if self.timing_tower is not None:
                self.timing_tower[1].record(main)
            if stamp is not None:            # (in front of the fork: the edge's cost on this queue is not the launch's)
                L.check(lib.dctr_stamp(stamp[1], mh), "dctr_stamp")
            if side is not None:
                side.wait_stream(main)       # the update may start once the first launch is done
            L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(b.desc), _ptr(b.out), ld, B, _ptr(b.g_logit), _ptr(b.ws),
                                             _ptr(loss), _ptr(g_bias), ctypes.byref(inline), mh), "dctr_mlp_train_wgrad")
            with on_side():
                sh = L.stream_handle(dev)
                if self.timing is not None:
                    self.timing[0].record(side if side is not None else main)
                if stamp is not None:
                    L.check(lib.dctr_stamp(stamp[2], sh), "dctr_stamp")
                L.check(lib.dctr_embed_update(cplan, units, n_units, plan.max_vocab, _ptr(b.ids_t), _ptr(b.parts_t), B,
                                              _ptr(b.gx), ld, _ptr(b.out), ld, _ptr(b.fm_s), ld_s,
                                              _ptr(b.g_logit) if self.want_fm else None,
                                              _ptr(b.g_logit) if plan.has_wide else None, 1, opt, lr, eps, _ptr(xb),
                                              xb.stride(0), _ptr(g_wd), ctypes.byref(inline) if g_wd is not None else None,
                                              _ptr(ws_u), b.upd_n, 1, sh), "dctr_embed_update")
                if stamp is not None:
                    L.check(lib.dctr_stamp(stamp[3], sh), "dctr_stamp")
                if self.timing is not None:
                    self.timing[1].record(side if side is not None else main)
            if side is not None:
                # the main queue waits for the UPDATE (an event at the side queue's tail of now); what follows on the side
                # queue -- the next batch's pre-pass -- is not waited for
                ev = self._join_event()
                ev.record(side)
                nxt = next_xb if (next_xb is not None and torch.cuda.is_current_stream_capturing() and
                                  tuple(next_xb.shape) == tuple(xb.shape) and next_xb.dtype == xb.dtype and
                                  next_xb.stride() == xb.stride()) else None
                if nxt is not None:
                    with on_side():
                        self._prepass(b, cplan, nxt, B, L.stream_handle(dev))
                    self._prepassed = (nxt.data_ptr(), nxt.stride(0), B, id(b), True)
                main.wait_event(ev)
        finally:
            slab.inline_done = True
            slab.end_inline_step()
        return loss, y_pred

    # ---- round 6: tower -> update -> tower on one queue, the weights handed over through a word in memory --------------
    @staticmethod
    def topology():
        """update_side (default): the round-4 two-queue step.  weights_flag (round 6, opt-in): tower -> update on ONE queue,
        the dense parameters handed to the next tower launch through a word in memory (_step_flag) -- bit-identical
        (tests/test_gpu_step_engine.py) and measured SLOWER, 0.107 against 0.090 ms per step: every cross-queue edge of a
        hipGraph costs the queues it touches 6-12 us, and this arrangement still has three of them per step on the main queue
        (profiles/r06_step_edges.txt; DESIGN.md section 3 has the whole account, including the edge-free two-graph variant
        that was built, measured at 0.104 ms and removed)."""
        t = os.environ.get("DCTR_STEP_TOPOLOGY", "update_side")
        return t if t in ("weights_flag", "update_side", "serial") else "update_side"

    def _event(self):
        """events that live as long as the engine, handed out round-robin (see _join_event)"""
        if self._pool is None:
            self._pool = [torch.cuda.Event() for _ in range(12)]
        self._pool_i = (self._pool_i + 1) % len(self._pool)
        return self._pool[self._pool_i]

    def _step_flag(self, xb, yb, mode, next_xb=None):
        lib = L.lib()
        model, slab = self.model, self.slab
        plan = model.model_plan()
        dev = xb.device
        B = xb.shape[0]
        b = self._buffers(B, dev)
        self.steps_run += 1
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            b.pinned = True
        a = b if self._parity == 0 else b.alt      # the activation set of this step
        self._parity ^= 1
        cplan = plan.bind(dev)
        y = yb.reshape(-1)
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.float().contiguous()
        y_pred = torch.empty((B,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        kind = plan.update[0]
        opt = L.UPD_ADAGRAD if kind == "adagrad" else L.UPD_SGD
        lr = float(plan.update[1])
        eps = float(plan.update[2]) if kind == "adagrad" else 0.0
        if not slab.begin_inline_step(mode[0], mode[1], mode[2] if len(mode) > 2 else 0.0):
            raise RuntimeError("the gather step needs a plain SGD / Adagrad dense optimizer")
        inline = slab.inline
        bias = model.out.bias
        g_bias = slab.grad_of(bias)
        lw = plan.wide_dense_weight
        g_wd = slab.grad_of(lw) if lw is not None else None
        err = plan.err_flag(dev)
        ws_u = b.upd_ws
        if getattr(ws_u, "_dctr_owner", None) is not None:      # an abandoned pre-pass of the autograd route left counts
            ws_u.zero_()
            ws_u._dctr_owner = None
            self._prepassed = None
        if self._sync is None or self._sync.device != dev:
            self._sync = torch.zeros((L.SYNC_INTS,), dtype=torch.int32, device=dev)
        units, n_units = plan.units_ptr(), plan.n_grid_units
        plan.point_step_buffers(b.den_t, b.amax)
        ld = plan.ld_out
        ld_s = a.fm_s.stride(0) if a.fm_s is not None else 0
        main = torch.cuda.current_stream(dev)
        side = _streams.side_stream(dev, "seg")
        token = (xb.data_ptr(), xb.stride(0), B, id(b), bool(capturing))
        have_prepass = self._prepassed is not None and self._prepassed == token and token[4]
        self._prepassed = None
        timeout_us = int(os.environ.get("DCTR_STEP_WAIT_US", "2000000"))
        dbg = int(os.environ.get("DCTR_DBG_EDGES", "0"))     # TIMING EXPERIMENTS ONLY: drop graph edges (wrong results)
        ok = False
        try:
            if not have_prepass:
                side.wait_stream(main)       # X is complete; (first step of a capture: the side queue joins the capture)
                with torch.cuda.stream(side):
                    self._prepass(b, cplan, xb, B, L.stream_handle(dev))
            ev_pre = self._event()
            ev_pre.record(side)              # this batch's pre-pass (enqueued here or by the previous step) is done
            mh = L.stream_handle(dev)
            if self.timing_tower is not None:
                self.timing_tower[0].record(main)
            L.check(lib.dctr_embed_tower_train_step_sync(
                cplan, _ptr(xb), xb.stride(0), ctypes.byref(a.desc), B, 1 if self.want_fm else 0, _ptr(bias), _ptr(y),
                _ptr(y_pred), _ptr(a.g_logit), _ptr(a.gx), ld, _ptr(a.out), ld, _ptr(a.fm_s), ld_s, _ptr(err), _ptr(a.ws),
                _ptr(self._sync), timeout_us, mh), "dctr_embed_tower_train_step_sync")
            if self.timing_tower is not None:
                self.timing_tower[1].record(main)
            ev_t = self._event()
            ev_t.record(main)
            # side queue: the weight gradients with their reduction and the dense optimizer step; the launch's last reducer
            # advances the weights' generation, which the NEXT tower launch waits for in its kernel
            if not (dbg & 4):
                side.wait_event(ev_t)
            with torch.cuda.stream(side):
                L.check(lib.dctr_mlp_train_wgrad_sync(ctypes.byref(a.desc), _ptr(a.out), ld, B, _ptr(a.g_logit), _ptr(a.ws),
                                                      _ptr(loss), _ptr(g_bias), ctypes.byref(inline), _ptr(self._sync),
                                                      _ptr(b.cnt), L.stream_handle(dev)), "dctr_mlp_train_wgrad_sync")
            # main queue: the update, right behind the tower (its pre-pass finished long ago: an edge with slack)
            if not (dbg & 2):
                main.wait_event(ev_pre)
            if self.timing is not None:
                self.timing[0].record(main)
            L.check(lib.dctr_embed_update(cplan, units, n_units, plan.max_vocab, _ptr(b.ids_t), _ptr(b.parts_t), B,
                                          _ptr(a.gx), ld, _ptr(a.out), ld, _ptr(a.fm_s), ld_s,
                                          _ptr(a.g_logit) if self.want_fm else None,
                                          _ptr(a.g_logit) if plan.has_wide else None, 1, opt, lr, eps, _ptr(xb),
                                          xb.stride(0), _ptr(g_wd), ctypes.byref(inline) if g_wd is not None else None,
                                          _ptr(ws_u), b.upd_n, 1, mh), "dctr_embed_update")
            if self.timing is not None:
                self.timing[1].record(main)
            nxt = next_xb if (next_xb is not None and capturing and tuple(next_xb.shape) == tuple(xb.shape) and
                              next_xb.dtype == xb.dtype and next_xb.stride() == xb.stride()) else None
            if nxt is not None:
                # the next batch's pre-pass on the side queue, behind the weight gradients and behind THIS update (which
                # reads the buffers the pre-pass rewrites)
                if not (dbg & 1):
                    ev_u = self._event()
                    ev_u.record(main)
                    side.wait_event(ev_u)
                with torch.cuda.stream(side):
                    self._prepass(b, cplan, nxt, B, L.stream_handle(dev))
                self._prepassed = (nxt.data_ptr(), nxt.stride(0), B, id(b), True)
            else:
                # the last step of a captured group, or an eager step: whatever follows on the caller's queue (the loss, a
                # predict, another model) sees the dense parameters stepped
                main.wait_stream(side)
            ok = True
        finally:
            slab.inline_done = True
            slab.end_inline_step()
            if not ok and not capturing:
                # a launch of the pair may be missing: the generations could disagree from here on -- start over
                torch.cuda.synchronize(dev)
                self._sync.zero_()
                b.cnt.zero_()
        return loss, y_pred

    def _join_event(self):
        """Two events that live as long as the engine, used alternately (an event created inside a hipGraph capture and
        collected during a later one aborts the process: graph.no_gc_during_capture)."""
        if self._events is None:
            self._events = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev_i = 1 - self._ev_i
        return self._events[self._ev_i]
