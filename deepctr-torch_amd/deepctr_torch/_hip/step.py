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
    """Everything one batch size needs, allocated once: activations, gradients, workspaces, the tower descriptor."""
    __slots__ = ("B", "out", "gx", "fm_s", "g_logit", "hs", "dhs", "ws", "ids_t", "parts_t", "desc", "upd_ws", "upd_n",
                 "keep")


class GatherStep(object):
    """See the module docstring.  Built by ``BaseModel._fused_step_state`` for models that declare
    ``_gather_step = True`` (their ``logit_parts`` is ``[linear, (fm), tower]`` over one fused lookup)."""

    def __init__(self, model, slab):
        self.model, self.slab = model, slab
        self._bufs = {}
        spec = _mlp.tower_layers(model.dnn, model.dnn_linear)
        self.layers, self.w_out = spec
        self.want_fm = bool(getattr(model, "use_fm", False)) and len(model.model_plan().deep) > 0
        self.sync = None          # topology "fused_flags": the device-side dependency words (include/dctr.h)
        self.timing = None        # (start, end) events around the update on its queue: bench.py's in-step roofline

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
        b.ids_t = torch.empty((len(plan.units), B), dtype=torch.int32, device=dev)
        b.parts_t = torch.empty((len(plan.units), B), dtype=torch.int16, device=dev)
        b.upd_ws, b.upd_n = plan.update_workspace(B, dev, always=True)
        b.keep = (Ws, gWs, gbs, g_wo)
        if len(self._bufs) >= 8:
            self._bufs.pop(next(iter(self._bufs)))
        self._bufs[key] = b
        return b

    # ---- one step -------------------------------------------------------------------------------------------------
    def step(self, xb, yb, mode, defer_join=False):
        """Enqueue one train step on (xb, yb); returns (loss, y_pred) device tensors.  ``mode``: the dense optimizer's
        (kind, lr, eps).  ``defer_join``: inside a multi-step hipGraph capture (not its last step) with the "fused_flags"
        topology the main queue does not wait for the side queue through a graph edge."""
        lib = L.lib()
        model, slab = self.model, self.slab
        plan = model.model_plan()
        dev = xb.device
        cuda = dev.type == "cuda"
        B = xb.shape[0]
        b = self._buffers(B, dev)
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
        units, n_units = plan.units_ptr(), len(plan.units)
        ld = plan.ld_out
        ld_s = b.fm_s.stride(0) if b.fm_s is not None else 0
        topo = os.environ.get("DCTR_STEP_TOPOLOGY", "update_side")
        serial = topo == "serial" or not cuda
        flags = topo == "fused_flags" and cuda
        if flags and self.sync is None and torch.cuda.is_current_stream_capturing():
            flags = False        # (a capture without an eager step in front of it: plain graph edges)
        elif flags:
            self._sync_block(dev)
        main = torch.cuda.current_stream(dev) if cuda else None
        side = _streams.side_stream(dev, "seg") if (cuda and not serial) else None
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        try:
            if side is not None:
                side.wait_stream(main)       # X is complete; (first step of a capture: the side queue joins the capture)
            with on_side():
                sh = L.stream_handle(dev)
                L.check(lib.dctr_embed_ids(cplan, units, n_units, _ptr(xb), xb.stride(0), B, _ptr(b.ids_t),
                                           _ptr(b.parts_t), sh), "dctr_embed_ids")
                L.check(lib.dctr_embed_segments(cplan, units, n_units, plan.max_vocab, _ptr(b.ids_t), _ptr(b.parts_t), B,
                                                _ptr(ws_u), b.upd_n, sh), "dctr_embed_segments")
            mh = L.stream_handle(dev)
            L.check(lib.dctr_embed_tower_train_step(cplan, _ptr(xb), xb.stride(0), ctypes.byref(b.desc), B,
                                                    1 if self.want_fm else 0, _ptr(bias), _ptr(y), _ptr(y_pred),
                                                    _ptr(b.g_logit), _ptr(b.gx), ld, _ptr(b.out), ld, _ptr(b.fm_s), ld_s,
                                                    _ptr(err), _ptr(b.ws), mh), "dctr_embed_tower_train_step")
            if side is not None:
                side.wait_stream(main)       # the update may start once the first launch is done
            L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(b.desc), _ptr(b.out), ld, B, _ptr(b.g_logit), _ptr(b.ws),
                                             _ptr(loss), _ptr(g_bias), ctypes.byref(inline), mh), "dctr_mlp_train_wgrad")
            with on_side():
                sh = L.stream_handle(dev)
                if self.timing is not None:
                    self.timing[0].record(side if side is not None else main)
                L.check(lib.dctr_embed_update(cplan, units, n_units, plan.max_vocab, _ptr(b.ids_t), _ptr(b.parts_t), B,
                                              _ptr(b.gx), ld, _ptr(b.out), ld, _ptr(b.fm_s), ld_s,
                                              _ptr(b.g_logit) if self.want_fm else None,
                                              _ptr(b.g_logit) if plan.has_wide else None, 1, opt, lr, eps, _ptr(xb),
                                              xb.stride(0), _ptr(g_wd), ctypes.byref(inline) if g_wd is not None else None,
                                              _ptr(ws_u), b.upd_n, 1, sh), "dctr_embed_update")
                if self.timing is not None:
                    self.timing[1].record(side if side is not None else main)
                if flags and defer_join:
                    L.check(lib.dctr_step_signal(_ptr(self._sync_block(dev)), L.SYNC_UPDATE, sh), "dctr_step_signal")
            if side is not None:
                if flags and defer_join:
                    # no graph edge from the side queue back to the main one: a one-wave kernel on the main queue polls
                    # the word the side queue's signal kernel advances behind the update
                    L.check(lib.dctr_step_wait(_ptr(self._sync_block(dev)), L.SYNC_UPDATE, 20000, mh), "dctr_step_wait")
                else:
                    main.wait_stream(side)
        finally:
            slab.inline_done = True
            slab.end_inline_step()
        return loss, y_pred

    def _sync_block(self, dev):
        if self.sync is None or self.sync.device != dev:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the step's sync block must exist before a hipGraph capture begins (run one eager step)")
            self.sync = torch.zeros(L.SYNC_INTS, dtype=torch.int32, device=dev)
        return self.sync
