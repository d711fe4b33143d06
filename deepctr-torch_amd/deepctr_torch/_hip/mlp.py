"""The DNN tower (+ ``dnn_linear``) and the BCE prediction head on the C-ABI kernels of csrc/mlp.hip / head.hip.

``tower(dnn, dnn_linear, x)`` == ``dnn_linear(dnn(x))`` of the reference (layers/core.py:120-134 followed by
deepfm.py:84) whenever the tower is relu / linear without BatchNorm and with inactive dropout -- the only
configuration the BASELINE models use.  Anything else (BatchNorm, Dice, PReLU, active dropout) keeps the module
path on PyTorch-ROCm; that is a different GPU implementation, not a CPU fallback.

Gradients take one of two routes:
  * autograd route (default): ``TowerFunction.backward`` returns dW / dbias / dw_out like any Function;
  * sink route (fused train step, ``dense.DenseSlab``): the kernels write the gradients straight into the flat
    gradient slab the fused dense optimizer consumes and autograd sees ``None``.
"""
import ctypes

import os
import threading

import torch
import torch.nn as nn

from . import lib as L
from ..layers.activation import Identity

MAX_TOWER_WIDTH = 1152


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _r4(n):
    return (int(n) + 3) // 4 * 4


def tower_layers(dnn, dnn_linear=None):
    """[(weight, bias, relu)], w_out  -- or None when the module combination is outside the kernels."""
    if dnn is None or getattr(dnn, "use_bn", False):
        return None
    if dnn.dropout_rate and dnn.training:
        return None
    if len(dnn.linears) > L.MLP_MAX_LAYERS:
        return None
    layers = []
    for fc, act in zip(dnn.linears, dnn.activation_layers):
        if type(act) is nn.ReLU:
            relu = 1
        elif type(act) is Identity:
            relu = 0
        else:
            return None
        # csrc/mlp.hip keeps a 16-sample tile of the widest activation (twice) next to a chunk of staged input columns
        # in LDS; the chunk shrinks from 512 to 64 columns as the tower widens (pick_kc), and the backward's two
        # gradient tiles 16 * 2 * (round64(width) + 4) * 4 bytes <= 150 KB bound the width at 1152 -- 1024-wide towers
        # run on the kernels.  Wider ones stay on PyTorch-ROCm's nn.Linear.
        if fc.out_features > MAX_TOWER_WIDTH or fc.weight.dtype != torch.float32:
            return None
        layers.append((fc.weight, fc.bias, relu))
    w_out = None
    if dnn_linear is not None:
        if dnn_linear.bias is not None or dnn_linear.out_features != 1 or \
                dnn_linear.in_features != dnn.linears[-1].out_features:
            return None
        w_out = dnn_linear.weight
    return layers, w_out


def _rows4(W):
    """(tensor usable as a [N, K] weight with a 16-byte aligned base and ld % 4 == 0, ld).  A slab-seated
    parameter (dense.DenseSlab) already is; anything else gets a zero-padded copy."""
    if W.dim() == 2 and (W.shape[1] == 1 or W.stride(1) == 1) and W.stride(0) % 4 == 0 and W.stride(0) >= W.shape[1] \
            and W.data_ptr() % 16 == 0:
        return W, W.stride(0)
    ld = _r4(W.shape[1])
    buf = torch.zeros((W.shape[0], ld), dtype=torch.float32, device=W.device)
    buf[:, :W.shape[1]].copy_(W.detach())
    return buf, ld


class _Meta(object):
    """Static description of one tower call (kept out of autograd's tensor arguments)."""

    def __init__(self, relus, has_out, K, sink=None, param_refs=None, keep=False):
        self.relus, self.has_out, self.K, self.sink = list(relus), bool(has_out), int(K), sink
        self.param_refs = param_refs     # the nn.Parameters, for the sink route
        self.keep = bool(keep)           # a backward can follow (grad mode is off inside Function.forward)


def _fill(desc, meta, Ws, lds, biases, hs, dhs, gWs, gbs, w_out, g_w_out):
    desc.n_layers = len(Ws)
    K = meta.K
    for l, W in enumerate(Ws):
        e = desc.layer[l]
        e.W = W.data_ptr()
        e.bias = biases[l].data_ptr() if biases[l] is not None else None
        e.h = hs[l].data_ptr() if hs[l] is not None else None
        e.dh = dhs[l].data_ptr() if dhs is not None else None
        e.gW = gWs[l].data_ptr() if gWs is not None and gWs[l] is not None else None
        e.gbias = gbs[l].data_ptr() if gbs is not None and gbs[l] is not None else None
        e.K, e.N, e.ld_w = K, W.shape[0], lds[l]
        e.ld_h = hs[l].stride(0) if hs[l] is not None else _r4(W.shape[0])
        e.relu = meta.relus[l]
        K = W.shape[0]
    desc.w_out = w_out.data_ptr() if w_out is not None else None
    desc.g_w_out = g_w_out.data_ptr() if g_w_out is not None else None


class TowerFunction(torch.autograd.Function):
    """x [B, ld_x] (first ``meta.K`` columns are the tower input) -> logit [B, 1] (with w_out) or h_L [B, N_L]."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        lib = L.lib()
        L.require_gpu(x, "DNN input")
        n = len(meta.relus)
        ctx.x_cols = x.shape[1]
        Wp = [params[2 * l] for l in range(n)]
        bp = [params[2 * l + 1] for l in range(n)]
        w_out = params[2 * n] if meta.has_out else None
        if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 \
                or x.stride(0) < meta.K:
            buf = torch.zeros((x.shape[0], _r4(meta.K)), dtype=torch.float32, device=x.device)
            buf[:, :meta.K].copy_(x[:, :meta.K])
            x = buf
        B = x.shape[0]
        keep = meta.keep
        Ws, lds = [], []
        for W in Wp:
            w, ld = _rows4(W)
            Ws.append(w)
            lds.append(ld)
        hs = []
        for l, W in enumerate(Wp):
            need = keep or (l == n - 1 and not meta.has_out)
            hs.append(torch.empty((B, _r4(W.shape[0])), dtype=torch.float32, device=x.device) if need else None)
        logit = torch.empty((B,), dtype=torch.float32, device=x.device) if meta.has_out else None
        wo = w_out.reshape(-1) if w_out is not None else None
        if wo is not None and not wo.is_contiguous():
            wo = wo.contiguous()
        desc = L.Mlp()
        _fill(desc, meta, Ws, lds, bp, hs, None, None, None, wo, None)
        L.check(lib.dctr_mlp_fwd(ctypes.byref(desc), _ptr(x), x.stride(0), B, _ptr(logit), L.stream_handle(x.device)),
                "dctr_mlp_fwd")
        ctx.meta, ctx.n = meta, n
        ctx.lds = lds
        ctx.padded = [w is not W for w, W in zip(Ws, Wp)]
        if keep:
            ctx.save_for_backward(x, wo, *(Ws + bp + hs))
        ctx.set_materialize_grads(False)
        if meta.has_out:
            return logit.unsqueeze(1)
        return hs[-1][:, :Wp[-1].shape[0]]

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        meta, n = ctx.meta, ctx.n
        saved = ctx.saved_tensors
        x, wo = saved[0], saved[1]
        Ws, bp, hs = list(saved[2:2 + n]), list(saved[2 + n:2 + 2 * n]), list(saved[2 + 2 * n:2 + 3 * n])
        n_ret = 2 + 2 * n + (1 if meta.has_out else 0)
        if g is None:
            return (None,) * n_ret
        B, dev = x.shape[0], x.device
        if meta.has_out:
            g = g.reshape(B)
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            ld_g = 0
        else:
            if g.dtype != torch.float32 or g.stride(1) != 1:
                g = g.float().contiguous()
            ld_g = g.stride(0)
        sink = meta.sink
        dhs = [torch.empty_like(h) for h in hs]
        need_gx = ctx.needs_input_grad[0]
        gx = torch.empty((B, x.stride(0)), dtype=torch.float32, device=dev) if need_gx else None
        gWs, gbs, rets = [], [], []
        for l in range(n):
            W, b = Ws[l], bp[l]
            N, Kl, ld = W.shape[0], (meta.K if l == 0 else Ws[l - 1].shape[0]), ctx.lds[l]
            gw_sink = sink.grad_of(meta.param_refs[2 * l]) if sink is not None else None
            gb_sink = sink.grad_of(meta.param_refs[2 * l + 1]) if (sink is not None and b is not None) else None
            if gw_sink is not None and (gw_sink.dim() != 2 or gw_sink.stride(0) != ld):
                raise RuntimeError("dense slab and tower disagree on a weight's leading dimension")
            gW = gw_sink if gw_sink is not None else torch.empty((N, ld), dtype=torch.float32, device=dev)
            gb = None
            if b is not None:
                gb = gb_sink if gb_sink is not None else torch.empty((N,), dtype=torch.float32, device=dev)
            gWs.append(gW)
            gbs.append(gb)
            rets.append(None if gw_sink is not None else (gW if ld == Kl else gW[:, :Kl]))
            rets.append(None if (b is None or gb_sink is not None) else gb)
        g_wo = None
        if meta.has_out:
            go_sink = sink.grad_of(meta.param_refs[2 * n]) if sink is not None else None
            if go_sink is not None:
                go_sink = go_sink.reshape(-1)
            g_wo = go_sink if go_sink is not None else torch.empty((wo.shape[0],), dtype=torch.float32, device=dev)
            rets.append(None if go_sink is not None else g_wo.reshape(1, -1))
        desc = L.Mlp()
        _fill(desc, meta, Ws, ctx.lds, bp, hs, dhs, gWs, gbs, wo, g_wo)
        ws = torch.empty((max(1, lib.dctr_mlp_bwd_workspace_floats(ctypes.byref(desc), B)),), dtype=torch.float32,
                         device=dev)
        L.check(lib.dctr_mlp_bwd(ctypes.byref(desc), _ptr(x), x.stride(0), B, _ptr(g), ld_g, _ptr(gx),
                                 gx.stride(0) if gx is not None else 0, _ptr(ws), L.stream_handle(dev)), "dctr_mlp_bwd")
        if gx is not None and gx.shape[1] != ctx.x_cols:
            gx = gx[:, :ctx.x_cols]
        return (gx, None) + tuple(rets)


class _TunedGemm(object):
    """PyTorch-ROCm's TunableOp (the search over hipBLASLt / rocBLAS solutions) around the plain library GEMMs of a very
    wide layer's BACKWARD: the library's default picks for [4096 x 128] x [128 x 10 413] and [128 x 4096] x [4096 x 10 413]
    run at 70-83 TFLOP/s, the tuned ones at ~105 (profiles/r04_fibinet_tunableop.txt).  Scoped: the process-wide switches are restored on exit; nothing is
    written to the working directory (the picks go to a file in the temp directory unless the user named one); inside a hipGraph capture only cached picks are used
    (a search launches and times kernels).

    What it costs (round-4 advisor): the switches are PROCESS-wide -- a GEMM another thread issues while this block is open is
    tuned too, and two threads opening it would restore each other's state, hence the lock held for the block's duration --
    and the search is timing-based: two runs may settle on different solutions for the two backward GEMMs, i.e. different
    summation orders in the last bits of FiBiNET-sized layers' gradients (every other kernel of this package has a fixed
    order).  Hence OFF by default since round 6 (the library's default picks: run-to-run identical bits, like every other
    kernel of this package; under torchrun every rank then also picks alike -- round-5 advisor finding); ``DCTR_TUNABLE_GEMM=1``
    switches the search on (~5 % faster FiBiNET steps), ``PYTORCH_TUNABLEOP_FILENAME`` pins the picks of an earlier run."""
    named = False
    _lock = threading.RLock()

    def __enter__(self):
        _TunedGemm._lock.acquire()
        t = torch.cuda.tunable
        self.prev = (t.is_enabled(), t.tuning_is_enabled())
        if os.environ.get("DCTR_TUNABLE_GEMM", "0") != "1":
            return self
        if not self.prev[0] and not os.environ.get("PYTORCH_TUNABLEOP_FILENAME") and not _TunedGemm.named:
            # (this torch appends every pick to the results file as it is found: keep it out of the working directory)
            import tempfile
            t.set_filename(os.path.join(tempfile.gettempdir(), "dctr_tunableop_%d.csv" % os.getuid()), True)
            _TunedGemm.named = True
        t.enable(True)
        t.tuning_enable(not torch.cuda.is_current_stream_capturing())
        return self

    def __exit__(self, *exc):
        try:
            t = torch.cuda.tunable
            t.tuning_enable(self.prev[1])
            t.enable(self.prev[0])
        finally:
            _TunedGemm._lock.release()
        return False


_PADDED_W = {}


def _padded_weight(W, ld):
    """[N, ld] copy of ``W [N, K]`` with zero columns behind K, in a buffer kept per (weight storage, ld): the pad is written
    once, the K columns at every call (the weight moves every step)."""
    key = (W.data_ptr(), int(ld), tuple(W.shape), str(W.device))
    buf = _PADDED_W.get(key)
    if buf is None:
        if len(_PADDED_W) > 8:
            _PADDED_W.clear()
        buf = _PADDED_W[key] = torch.zeros((W.shape[0], int(ld)), dtype=W.dtype, device=W.device)
    buf[:, :W.shape[1]].copy_(W.detach())
    return buf


def _act_backward(g, h, relu, want_bias):
    """(``g`` through relu's backward as a contiguous tensor, the bias gradient or None): one pass over ``g``
    (csrc/head.hip k_colsum_part + the fixed-order finish) instead of threshold_backward + a two-stage ATen sum(0)."""
    gb = None
    if g.is_cuda and g.dtype == torch.float32 and want_bias and g.dim() == 2 and \
            g.stride(1) == 1 and os.environ.get("DCTR_GLUE_KERNELS", "1") != "0":
        lib = L.lib()
        B, N = g.shape
        go = torch.empty((B, N), dtype=torch.float32, device=g.device) if relu else None
        gb = torch.empty((N,), dtype=torch.float32, device=g.device)
        ws = torch.empty((max(1, lib.dctr_relu_bwd_bias_workspace_floats(B, N)),), dtype=torch.float32, device=g.device)
        L.check(lib.dctr_relu_bwd_bias(_ptr(g), g.stride(0), _ptr(h) if relu else None,
                                       h.stride(0) if relu else 0, B, N, _ptr(go), N, _ptr(gb), _ptr(ws),
                                       L.stream_handle(g.device)), "dctr_relu_bwd_bias")
        if relu:
            g = go
        elif not g.is_contiguous():
            g = g.contiguous()
    elif relu:
        g = torch.ops.aten.threshold_backward(g, h, 0)     # (relu's own backward: one launch)
    elif not g.is_contiguous():
        g = g.contiguous()
    return g, gb


class WideLinearFunction(torch.autograd.Function):
    """act(x[:, :K] W^T + b) for a layer too wide for the tower kernels (K > 4096): three library GEMMs (forward, input
    gradient, weight gradient), the two backward ones under ``_TunedGemm``; the relu mask and the bias gradient without
    autograd's extra nodes."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        # (the forward keeps the library's default pick: the faster ones TunableOp finds for K = 10 413 accumulate in longer
        # chains -- rms error 1.1e-6 against 0.8e-6 -- and the full-size gradient fixtures then sit at 0.6-1.3 of their bars
        # instead of 0.5; tools/probes/tuned_gemm_error.py.  The two backward GEMMs lose nothing: K = 128 and K = B.)
        # (relu in the GEMM's epilogue -- torch._addmm_activation -- and the bias gradient as a matrix-vector product were
        # measured: 0.848 against 0.833 ms per FiBiNET step, the library then picks a slower solution)
        h = torch.addmm(b, x, W.t()) if b is not None else torch.mm(x, W.t())
        if relu:
            h = torch.relu_(h)
        ctx.relu, ctx.has_bias = bool(relu), b is not None
        ctx.save_for_backward(x, W, h if relu else None)
        return h

    @staticmethod
    def backward(ctx, g):
        x, W, h = ctx.saved_tensors
        g, gb = _act_backward(g, h, ctx.relu, ctx.has_bias and ctx.needs_input_grad[2])
        gx = gW = None
        with _TunedGemm():
            if ctx.needs_input_grad[0]:
                if x.stride(0) != x.shape[1]:
                    # the input came with padded rows (ops.slab_ld: rows on 128-byte lines): its gradient goes back with the
                    # same row stride -- the kernels that consume it read 64-byte pieces per (sample, pair)
                    # (through a zero-padded copy of W: with `out=` a strided view the library picked a 24 us slower GEMM)
                    Wp = _padded_weight(W, x.stride(0))
                    gx = torch.mm(g, Wp)[:, :W.shape[1]]
                else:
                    gx = torch.mm(g, W)
            if ctx.needs_input_grad[1]:
                gW = torch.mm(g.t(), x)
        if gb is None and ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gW, gb, None


class PendingPairs(object):
    """FiBiNET's DNN input ``[Bilinear(senet) | Bilinear(raw) | dense]`` (fibinet.py:82-87) NOT yet computed: what
    ``BilinearInteraction.fused_pair(..., lazy=True)`` hands to the tower.  ``tower`` then runs pairs and first layer as
    ONE autograd node (``BilinearWideFunction``: the gradient of the 170 MB product slab is never materialised) when the
    shapes fit csrc/bilinear_wide.hip, else ``materialize()`` gives the slab and everything runs as before."""

    def __init__(self, meta, raw, senet, dense, weights):
        self.meta, self.raw, self.senet, self.dense, self.weights = meta, raw, senet, dense, tuple(weights)
        B, F, D = raw.shape
        self.shape = (B, F * (F - 1) * D + (dense.shape[1] if dense is not None else 0))
        self.is_cuda, self.device, self.dtype = raw.is_cuda, raw.device, raw.dtype
        self.requires_grad = raw.requires_grad or senet.requires_grad or any(w.requires_grad for w in weights)

    def fits(self, W0, relu0):
        F, D = self.raw.shape[1], self.raw.shape[2]
        return (self.is_cuda and os.environ.get("DCTR_BILINEAR_WIDE", "1") != "0" and D == 16 and
                self.meta.n_w == F * (F - 1) // 2 and W0.shape[0] <= 128 and W0.shape[0] % 4 == 0 and
                W0.shape[1] == self.shape[1] and W0.dtype == torch.float32 and W0.stride(1) == 1 and
                self.raw.dtype == torch.float32 and
                F <= 64 and 4 * 16 * (F * D + 4) * 4 + 8 * 2176 + ((F * (F - 1) // 2) // 4 + 8) * 128 <= 158 * 1024)

    def materialize(self):
        from . import ops as _ops
        return _ops.BilinearFunction.apply(self.meta, self.raw, self.senet, self.dense, *self.weights)


class BilinearWideFunction(torch.autograd.Function):
    """``act(W0 [Bilinear(V) | Bilinear(E) | dense] + b0)`` as one node.  Forward: ``dctr_bilinear_wide_fwd`` -- the pairs go
    to the first layer's MFMAs from registers; the DNN input is written as a by-product for the backward
    (``DCTR_BILINEAR_WIDE_FWD=0``: pair kernel + library GEMM, as ``BilinearFunction`` + ``WideLinearFunction``).  Backward: the weight gradient's GEMM
    over the saved DNN input, and ``dctr_bilinear_wide_bwd`` -- gE, gV and the pair weights' gradients straight from
    ``gh`` and ``W0`` on the matrix cores; no ``gh W0`` GEMM, no gradient slab, no second and third pass over it."""

    @staticmethod
    def forward(ctx, meta, relu, E, V, dense, W0, b0, *weights):
        from . import ops as _ops
        lib = L.lib()
        E, lde = _ops._rows3(E, "Bilinear input")
        V, ldv = _ops._rows3(V, "Bilinear second input")
        B, F, D = E.shape
        Wf = meta.flat_weights(weights)
        P = F * (F - 1) // 2
        n_dense = dense.shape[1] if dense is not None else 0
        if dense is not None and (dense.stride(1) != 1 or dense.dtype != torch.float32):
            dense = dense.float().contiguous()
        width = 2 * P * D + n_dense
        ld_out = _ops.slab_ld(width)
        # (whole tiles of 32 rows: the fused forward stores its pairs unconditionally)
        x = torch.empty(((B + 31) // 32 * 32, ld_out), dtype=torch.float32, device=E.device)[:B, :width]
        sched = meta.device_tables(E.device)
        wpk_b = None
        if os.environ.get("DCTR_BILINEAR_WIDE_FWD", "1") != "0" and W0.stride(1) == 1 and ld_out % 4 == 0 and n_dense <= 32 and \
                (b0 is None or (b0.dtype == torch.float32 and b0.is_contiguous())):
            # pairs and first layer in one launch: the pairs feed the matrix cores from registers; x is a by-product
            # (the backward's weight-gradient GEMM reads it)
            h = torch.empty((B, W0.shape[0]), dtype=torch.float32, device=E.device)
            ws = torch.empty((lib.dctr_bilinear_wide_fwd_workspace_floats(B, P),), dtype=torch.float32, device=E.device)
            if any(ctx.needs_input_grad):
                # W0 in the backward's operand layout, by the same packing launch
                wpk_b = torch.empty((lib.dctr_bilinear_wide_pack_floats(P),), dtype=torch.float32, device=E.device)
            L.check(lib.dctr_bilinear_wide_fwd(_ptr(E), lde, _ptr(V), ldv, _ptr(Wf), _ptr(sched[2]), P, F, D, B,
                                               _ptr(dense), dense.stride(0) if dense is not None else 0, n_dense,
                                               _ptr(W0), W0.stride(0), W0.shape[0], _ptr(b0), int(bool(relu)), _ptr(x),
                                               ld_out, _ptr(h), h.stride(0), _ptr(ws), _ptr(wpk_b),
                                               L.stream_handle(E.device)), "dctr_bilinear_wide_fwd")
        else:
            L.check(lib.dctr_bilinear_fwd(_ptr(E), lde, _ptr(V), ldv, _ptr(Wf), _ptr(sched[2]), sched[2].shape[0], P, F, D,
                                          B, _ptr(x), ld_out, _ptr(dense), dense.stride(0) if dense is not None else 0,
                                          n_dense, 2 * P * D, L.stream_handle(E.device)), "dctr_bilinear_fwd")
            h = torch.addmm(b0, x, W0.t()) if b0 is not None else torch.mm(x, W0.t())
            if relu:
                h = torch.relu_(h)
        ctx.meta, ctx.relu, ctx.has_bias, ctx.n_w_in = meta, bool(relu), b0 is not None, len(weights)
        ctx.save_for_backward(E, V, Wf, x, W0, h if relu else None, wpk_b)
        return h

    @staticmethod
    def backward(ctx, g):
        from . import ops as _ops
        lib = L.lib()
        meta = ctx.meta
        E, V, Wf, x, W0, h, wpk_b = ctx.saved_tensors
        E, lde = _ops._rows3(E, "Bilinear input")
        V, ldv = _ops._rows3(V, "Bilinear second input")
        B, F, D = E.shape
        P = F * (F - 1) // 2
        dev = E.device
        g, gb = _act_backward(g, h, ctx.relu, ctx.has_bias and ctx.needs_input_grad[6])
        if g.stride(1) != 1 or g.stride(0) % 4 != 0 or g.data_ptr() % 16 != 0:
            g = g.contiguous()
        gW0 = None
        if ctx.needs_input_grad[5]:
            with _TunedGemm():
                gW0 = torch.mm(g.t(), x)
        if gb is None and ctx.has_bias and ctx.needs_input_grad[6]:
            gb = g.sum(0)
        gE = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        gV = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        gW = torch.empty((meta.n_w, D, D), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.dctr_bilinear_wide_bwd_workspace_floats(B, P),), dtype=torch.float32, device=dev)
        sched4, pair_w = meta.wide_tables(dev)
        L.check(lib.dctr_bilinear_wide_bwd(_ptr(E), lde, _ptr(V), ldv, _ptr(Wf), _ptr(sched4), sched4.shape[0],
                                           _ptr(pair_w), meta.n_w, P, F, D, B, _ptr(g), g.stride(0), _ptr(W0),
                                           W0.stride(0), W0.shape[0], _ptr(gE), _ptr(gV), _ptr(gW), _ptr(ws),
                                           _ptr(wpk_b), L.stream_handle(dev)), "dctr_bilinear_wide_bwd")
        return (None, None, gE, gV, None, gW0, gb) + tuple(gW[i] for i in range(ctx.n_w_in))


def tower(dnn, dnn_linear, x, K=None, sink=None):
    """``dnn_linear(dnn(x[:, :K]))`` (or ``dnn(x[:, :K])`` when ``dnn_linear`` is None)."""
    spec = tower_layers(dnn, dnn_linear)
    K = x.shape[1] if K is None else K
    pairs = x if isinstance(x, PendingPairs) else None
    if pairs is not None and not (spec is not None and K == x.shape[1] and K > 4096 and len(spec[0]) >= 2 and
                                  pairs.fits(spec[0][0][0], spec[0][0][2])):
        x, pairs = pairs.materialize(), None
    if pairs is not None:
        W0, b0, relu0 = spec[0][0]
        h0 = BilinearWideFunction.apply(pairs.meta, bool(relu0), pairs.raw, pairs.senet, pairs.dense, W0, b0,
                                        *pairs.weights)
        layers, w_out = spec[0][1:], spec[1]
        x, K = h0, W0.shape[0]
    elif spec is not None and x.is_cuda and K > 4096 and len(spec[0]) >= 2:
        # A very wide FIRST layer (FiBiNET: 10 413 inputs) stays one hipBLASLt GEMM each way; everything behind it -- the
        # remaining layers, dnn_linear, their backward and weight gradients -- runs on the tower kernels over its output
        # (round 4: those small layers were 6 hipBLASLt GEMMs of 8-27 us plus ~10 elementwise / reduce launches per step).
        W0, b0, relu0 = spec[0][0]
        h0 = WideLinearFunction.apply(x[:, :K] if K != x.shape[1] else x, W0, b0, bool(relu0))
        layers, w_out = spec[0][1:], spec[1]
        x, K = h0, W0.shape[0]
    elif spec is None or not x.is_cuda or K > 4096:   # (a single very wide layer, or modules outside the kernels)
        h = dnn(x[:, :K] if K != x.shape[1] else x)
        return dnn_linear(h) if dnn_linear is not None else h
    else:
        layers, w_out = spec
    params = []
    for (W, b, _) in layers:
        params += [W, b]
    if w_out is not None:
        params.append(w_out)
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params))
    meta = _Meta([r for (_, _, r) in layers], w_out is not None, K, sink, params, keep)
    return TowerFunction.apply(x, meta, *params)


class PendingTower(object):
    """What ``BaseModel.tower_logit`` returns while a fused train step is being assembled: the tower is not run yet,
    it will run together with the head and its own backward in ``tower_head`` (one launch per row tile)."""

    def __init__(self, dnn, dnn_linear, x, K, sink):
        self.dnn, self.dnn_linear, self.x, self.K, self.sink = dnn, dnn_linear, x, K, sink


class TowerHeadFunction(torch.autograd.Function):
    """loss, y_pred = BCE(sum)(sigmoid(part0 + part1 + tower(x) + bias), y) with EVERYTHING of the tower done in the
    forward call: tower forward, head, backward-data, weight gradients (written to the dense gradient slab).
    ``backward`` only hands out the stored input gradients; valid because the train step differentiates
    loss (+ regularisers that do not touch these inputs): the incoming gradient is exactly 1."""

    @staticmethod
    def forward(ctx, x, y, bias, meta, n_parts, *rest):
        lib = L.lib()
        parts, params = list(rest[:n_parts]), rest[n_parts:]
        n = len(meta.relus)
        Wp = [params[2 * l] for l in range(n)]
        bp = [params[2 * l + 1] for l in range(n)]
        w_out = params[2 * n]
        sink = meta.sink
        B, dev = x.shape[0], x.device
        ctx.x_cols = x.shape[1]
        if x.dtype != torch.float32 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or x.stride(0) < meta.K:
            buf = torch.zeros((B, _r4(meta.K)), dtype=torch.float32, device=dev)
            buf[:, :meta.K].copy_(x[:, :meta.K])
            x = buf
        Ws, lds = [], []
        for W in Wp:
            w, ld = _rows4(W)
            if w is not W:
                raise RuntimeError("the fused train step needs slab-seated tower weights")
            Ws.append(w)
            lds.append(ld)
        hs = [torch.empty((B, _r4(W.shape[0])), dtype=torch.float32, device=dev) for W in Wp]
        dhs = [torch.empty_like(h) for h in hs]
        gWs = [sink.grad_of(meta.param_refs[2 * l]) for l in range(n)]
        gbs = [sink.grad_of(meta.param_refs[2 * l + 1]) if bp[l] is not None else None for l in range(n)]
        g_wo = sink.grad_of(meta.param_refs[2 * n]).reshape(-1)
        g_bias = sink.grad_of(meta.bias_ref) if bias is not None else None
        ps = []
        for p in parts:
            q = p.reshape(-1)
            if q.dtype != torch.float32 or not q.is_contiguous():
                q = q.float().contiguous()
            ps.append(q)
        y = y.reshape(-1)
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.float().contiguous()
        wo = w_out.reshape(-1)
        y_pred = torch.empty((B,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        g_logit = torch.empty((B,), dtype=torch.float32, device=dev)
        gx = torch.empty((B, x.stride(0)), dtype=torch.float32, device=dev)
        desc = L.Mlp()
        _fill(desc, meta, Ws, lds, bp, hs, dhs, gWs, gbs, wo, g_wo)
        ws = torch.empty((max(1, lib.dctr_mlp_train_workspace_floats(ctypes.byref(desc), B)),), dtype=torch.float32,
                         device=dev)
        # topology "flags": the tower + head launch signals gx / g_logit complete in the step's sync block and the
        # update's stream waits for THAT (dctr_step_wait) instead of an event of this stream
        sync = None
        if getattr(sink, "flag_sync", False) and getattr(sink, "inline", None) is not None and \
                getattr(sink, "update_stream", None) is not None and x.device.type == "cuda":
            sync = sink.sync_block(dev)
            desc.step_sync = sync.data_ptr()
        pp = [_ptr(p) for p in ps] + [None] * (2 - len(ps))
        # The weight gradients need only what the first launch leaves behind (x, h, dh, g_logit) and nothing but the
        # dense optimizer needs THEM: with a fork stream on the sink they run beside the embedding update (which needs
        # only gx / g_logit) instead of in front of it.  The sink joins the fork before the dense optimizer step.
        inline = getattr(sink, "inline", None)
        if inline is not None and hasattr(sink, "join") and not getattr(sink, "gather_side", False):
            sink.join()       # a previous step's forked weight-gradient / optimizer kernels wrote the weights read below
        # (gather_side: the fork holds the embedding update and this step's gather -- the tower reads nothing they
        # write but the gathered rows, for which the gather's own launch made this stream wait)
        fork = sink.fork_stream(dev) if (hasattr(sink, "fork_stream") and inline is None) else None
        defer = fork is None and inline is None and getattr(sink, "overlap", False) == "defer" and x.device.type == "cuda"
        L.check(lib.dctr_mlp_train_step(ctypes.byref(desc), _ptr(x), x.stride(0), B, pp[0], pp[1], _ptr(bias), _ptr(y),
                                        _ptr(y_pred), _ptr(loss), _ptr(g_logit), _ptr(g_bias), _ptr(gx), gx.stride(0),
                                        _ptr(ws), 1 if (fork is not None or defer or inline is not None) else 0, None,
                                        L.stream_handle(dev)), "dctr_mlp_train_step")
        if inline is not None and getattr(sink, "wgrad_side", False) and x.device.type == "cuda":
            # Topology "tower_side": the weight gradients + their reduction (which also steps the parameters) go to
            # the fork stream, the embedding update stays on this one; nothing joins them until the NEXT tower launch
            # needs the stepped weights (DenseSlab.join() at the top of this function's next call, or at the end of the
            # step when the caller may read parameters).  Main chain: gather, tower, update, next gather.
            # The fork is ENQUEUED only after the embedding update has been (ops.EmbedFunction.backward calls
            # `sink.after_update`): hipGraph keeps a node's FIRST child on the node's own queue and sends later children
            # to other queues, whatever stream they were captured on -- launched here, ahead of the update, the weight
            # gradients stayed on the tower's queue and the update (the longer chain since round 3) paid a cross-queue
            # edge on both ends (~10 us each).
            # "tower_seg" (round 3): the fork IS the pre-pass's stream -- two queues in all (main: gather, tower, update;
            # side: ids, pre-pass, weight gradients + reduction), and the next tower launch waits for an event recorded
            # right behind the reduction, not for the side stream's tail (by then the next step's pre-pass).
            # tools/micro/topobench.hip (kernels that only busy-wait their body times) gives this recipe a period 4 us
            # shorter than "update_side".  The real step is SLOWER (0.107 ms against 0.097, profiles/
            # r03_step_topologies.json): the update and the weight gradients now start within a microsecond of each other
            # and k_mlp_wgrad takes 38-40 us beside the update's 1118 workgroups (22 us with the 6 us head start it has
            # in "update_side"), and that kernel is on this recipe's critical cycle.  Kept as an option, not the default.
            seg = getattr(sink, "update_stream", None) if getattr(sink, "wgrad_on_seg", False) else None
            side = seg if seg is not None else sink.fork_stream(dev, force=True)
            ev = sink.fork_event(0)
            ev.record(torch.cuda.current_stream(dev))       # fork point: right behind the tower kernel
            keep = (x, hs, dhs, ws, g_logit, loss, ps, y, wo, desc, inline)

            def launch_fork(side=side, ev=ev, keep=keep, B=B, g_bias=g_bias, by_event=seg is not None):
                x_, ws_, g_logit_, loss_, desc_, inline_ = keep[0], keep[3], keep[4], keep[5], keep[9], keep[10]
                side.wait_event(ev)
                L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(desc_), _ptr(x_), x_.stride(0), B, _ptr(g_logit_),
                                                 _ptr(ws_), _ptr(loss_), _ptr(g_bias), ctypes.byref(inline_),
                                                 ctypes.c_void_p(side.cuda_stream)), "dctr_mlp_train_wgrad")
                done = None
                if by_event:
                    done = sink.fork_event(1)
                    done.record(side)
                sink.forked(side, keep, done)
            sink.after_update = launch_fork
            sink.inline_done = True
            sink.update_stream = None            # (the update runs on this stream: ops.EmbedFunction.backward)
        elif inline is not None:
            # In-kernel optimizer: the weight gradients and their reduction follow in line on THIS stream and step the
            # parameters as they finish; the embedding update (which needs only gx / g_logit) is what leaves for the
            # side stream (ops.EmbedFunction.backward), right behind the event recorded here.  The step's critical
            # chain -- gather, tower, weight gradients -- then never crosses a queue (a cross-queue dependency costs
            # 6-10 us on this stack; round 1 paid two per step).
            upd = getattr(sink, "update_stream", None)
            if upd is not None and sync is not None:
                L.check(lib.dctr_step_wait(_ptr(sync), L.SYNC_TOWER, sink.sync_timeout_us,
                                           ctypes.c_void_p(upd.cuda_stream)), "dctr_step_wait(tower)")
            elif upd is not None:
                upd.wait_stream(torch.cuda.current_stream(dev))      # the update may start once this launch is done
            L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(desc), _ptr(x), x.stride(0), B, _ptr(g_logit), _ptr(ws),
                                             _ptr(loss), _ptr(g_bias), ctypes.byref(inline), L.stream_handle(dev)),
                    "dctr_mlp_train_wgrad")
            sink.inline_done = True
            if getattr(sink, "gather_side", False):
                # What the weight-gradient kernels above read stays allocated until the NEXT tower launch: the next
                # step's gather runs on the side stream, ordered behind this launch's first kernel only -- memory the
                # allocator handed from these tensors to the gather's outputs would be overwritten under the reader.
                sink.main_keep = (x, hs, dhs, ws, g_logit, loss, ps, y, wo, gx)
        if defer:
            keep = (x, hs, dhs, ws, g_logit, loss, ps, y, wo, gx, desc)

            def launch(stream, keep=keep, B=B, g_bias=g_bias):
                x_, ws_, g_logit_, loss_, desc_ = keep[0], keep[3], keep[4], keep[5], keep[10]
                L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(desc_), _ptr(x_), x_.stride(0), B, _ptr(g_logit_),
                                                 _ptr(ws_), _ptr(loss_), _ptr(g_bias), None,
                                                 ctypes.c_void_p(stream.cuda_stream)), "dctr_mlp_train_wgrad")
            sink.deferred = launch
        if fork is not None:
            side = fork
            side.wait_stream(torch.cuda.current_stream(dev))     # fork point: right behind the tower kernel
            L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(desc), _ptr(x), x.stride(0), B, _ptr(g_logit), _ptr(ws),
                                             _ptr(loss), _ptr(g_bias), None, ctypes.c_void_p(side.cuda_stream)),
                    "dctr_mlp_train_wgrad")
            # everything the forked kernels touch stays allocated until the join (no record_stream bookkeeping)
            sink.forked(side, (x, hs, dhs, ws, g_logit, loss, ps, y, wo))
        ctx.shapes = [tuple(p.shape) for p in parts]
        ctx.n_rest = len(rest)
        ctx.save_for_backward(gx, g_logit)
        ctx.mark_non_differentiable(y_pred)
        ctx.set_materialize_grads(False)
        return loss, y_pred

    @staticmethod
    def backward(ctx, g_loss, _g_pred):
        gx, g_logit = ctx.saved_tensors
        if g_loss is None:
            return (None,) * (5 + ctx.n_rest)
        if gx.shape[1] != ctx.x_cols:
            gx = gx[:, :ctx.x_cols]
        grads = [g_logit.view(s) for s in ctx.shapes]
        return (gx, None, None, None, None) + tuple(grads) + (None,) * (ctx.n_rest - len(grads))


def tower_head(pending, parts, bias, y):
    """The fused tower + head + loss (see TowerHeadFunction).  ``parts``: at most two other logit parts."""
    spec = tower_layers(pending.dnn, pending.dnn_linear)
    layers, w_out = spec
    params = []
    for (W, b, _) in layers:
        params += [W, b]
    params.append(w_out)
    meta = _Meta([r for (_, _, r) in layers], True, pending.K, pending.sink, params, True)
    meta.bias_ref = bias
    return TowerHeadFunction.apply(pending.x, y, bias, meta, len(parts), *(list(parts) + params))


def fusable_head(pending, parts):
    return (pending is not None and pending.sink is not None and len(parts) <= 2 and
            tower_layers(pending.dnn, pending.dnn_linear) is not None and pending.dnn_linear is not None and
            pending.x.is_cuda and pending.K <= 4096)


class BCEHeadFunction(torch.autograd.Function):
    """(loss, y_pred) = BCE(sum)(sigmoid(sum(parts) + bias), y) in ONE launch; the launch also produces
    d loss / d logit, which ``backward`` hands to every part (valid because the train step differentiates
    ``loss + regularisation + aux`` -- d total / d loss is exactly 1; ``backward`` still scales by the incoming
    gradient unless ``unit`` says it is known to be 1)."""

    @staticmethod
    def forward(ctx, y, bias, unit, g_bias_sink, *parts):
        lib = L.lib()
        ps = []
        for p in parts:
            L.require_gpu(p, "logit part")
            q = p.reshape(-1)
            if q.dtype != torch.float32 or not q.is_contiguous():
                q = q.float().contiguous()
            ps.append(q)
        B, dev = ps[0].shape[0], ps[0].device
        if len(ps) > 4:
            raise NotImplementedError("the fused head takes at most 4 logit parts")
        y = y.reshape(-1)
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.float().contiguous()
        y_pred = torch.empty((B,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        g_logit = torch.empty((B,), dtype=torch.float32, device=dev)
        g_bias = g_bias_sink if g_bias_sink is not None else (
            torch.empty((1,), dtype=torch.float32, device=dev) if bias is not None else None)
        pp = [_ptr(p) for p in ps] + [None] * (4 - len(ps))
        L.check(lib.dctr_bce_head(pp[0], pp[1], pp[2], pp[3], _ptr(bias), _ptr(y), B, _ptr(y_pred), _ptr(loss),
                                  _ptr(g_logit), _ptr(g_bias), L.stream_handle(dev)), "dctr_bce_head")
        ctx.unit, ctx.sunk = bool(unit), g_bias_sink is not None
        ctx.shapes = [tuple(p.shape) for p in parts]
        ctx.save_for_backward(g_logit, g_bias if (bias is not None and g_bias_sink is None) else None)
        ctx.mark_non_differentiable(y_pred)
        ctx.set_materialize_grads(False)
        return loss, y_pred

    @staticmethod
    def backward(ctx, g_loss, _g_pred):
        g_logit, g_bias = ctx.saved_tensors
        if g_loss is None:
            return (None,) * (4 + len(ctx.shapes))
        if not ctx.unit:
            g_logit = g_logit * g_loss
            if g_bias is not None:
                g_bias = g_bias * g_loss
        gb = None if (ctx.sunk or g_bias is None) else g_bias
        return (None, gb, None, None) + tuple(g_logit.view(s) for s in ctx.shapes)


def bce_head(parts, bias, y, unit=False, g_bias_sink=None):
    return BCEHeadFunction.apply(y, bias, unit, g_bias_sink, *parts)
