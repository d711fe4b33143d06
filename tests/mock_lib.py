"""A numpy stand-in for the tower / head / dense-optimizer entry points of libdctr_hip.so, for CPU tests of the
Python plumbing above the C-ABI (autograd Functions, DenseSlab, argument marshalling).  Test infrastructure only:
it decodes the very ctypes arguments the product code passes and computes with the oracle's formulas."""
import ctypes

import numpy as np

from np_oracle import sigmoid


def _arr(ptr, shape, ld=None, dtype=np.float32):
    """numpy view of host memory at ``ptr`` ([rows, cols] with leading dimension ld)."""
    if ptr is None:
        return None
    addr = ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)
    if not addr:
        return None
    if len(shape) == 1:
        n = shape[0]
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr))
    rows, cols = shape
    ld = cols if ld is None else int(ld)
    flat = np.ctypeslib.as_array((ctypes.c_float * (max(rows - 1, 0) * ld + cols)).from_address(addr))
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(ld * 4, 4))


class MockLib(object):
    def __init__(self):
        self.calls = []

    # ---- tower --------------------------------------------------------------------------------------------
    def _layers(self, mref):
        m = mref._obj
        out = []
        for l in range(m.n_layers):
            e = m.layer[l]
            assert e.ld_w % 4 == 0 and e.W % 16 == 0
            out.append(e)
        return m, out

    def dctr_mlp_fwd(self, mref, x, ld_x, B, logit, stream):
        self.calls.append("mlp_fwd")
        m, layers = self._layers(mref)
        h = _arr(x, (B, layers[0].K), ld_x).astype(np.float64)
        for e in layers:
            W = _arr(e.W, (e.N, e.K), e.ld_w).astype(np.float64)
            h = h @ W.T
            if e.bias:
                h = h + _arr(e.bias, (e.N,)).astype(np.float64)
            if e.relu:
                h = np.maximum(h, 0)
            if e.h:
                _arr(e.h, (B, e.N), e.ld_h)[...] = h
        if m.w_out:
            _arr(logit, (B,))[...] = h @ _arr(m.w_out, (layers[-1].N,)).astype(np.float64)
        return 0

    def dctr_mlp_bwd_workspace_floats(self, mref, B):
        return 16

    def dctr_mlp_bwd(self, mref, x, ld_x, B, g, ld_g, gx, ld_gx, ws, stream):
        self.calls.append("mlp_bwd")
        m, layers = self._layers(mref)
        top = layers[-1]
        if m.w_out:
            gh = np.outer(_arr(g, (B,)).astype(np.float64), _arr(m.w_out, (top.N,)).astype(np.float64))
            if m.g_w_out:
                _arr(m.g_w_out, (top.N,))[...] = _arr(g, (B,)).astype(np.float64) @ _arr(top.h, (B, top.N), top.ld_h)
        else:
            gh = _arr(g, (B, top.N), ld_g).astype(np.float64)
        for l in reversed(range(m.n_layers)):
            e = layers[l]
            if e.relu:
                gh = gh * (_arr(e.h, (B, e.N), e.ld_h) > 0)
            _arr(e.dh, (B, e.N), e.ld_h)[...] = gh
            inp = _arr(x, (B, e.K), ld_x) if l == 0 else _arr(layers[l - 1].h, (B, e.K), layers[l - 1].ld_h)
            if e.gW:
                full = _arr(e.gW, (e.N, e.ld_w), e.ld_w)
                full[...] = 0
                full[:, :e.K] = gh.T @ inp.astype(np.float64)
            if e.gbias:
                _arr(e.gbias, (e.N,))[...] = gh.sum(0)
            gh = gh @ _arr(e.W, (e.N, e.K), e.ld_w).astype(np.float64)
        if gx:
            _arr(gx, (B, layers[0].K), ld_gx)[...] = gh
        return 0

    def dctr_mlp_train_workspace_floats(self, mref, B):
        return 16

    def dctr_mlp_train_wgrad(self, mref, x, ld_x, B, g_logit, ws, loss, g_bias, stream):
        return 0    # the mock's train step has already produced the weight gradients

    def dctr_mlp_train_step(self, mref, x, ld_x, B, p0, p1, bias, y, y_pred, loss, g_logit, g_bias, gx, ld_gx, ws,
                            defer_wgrad, stream):
        """forward + head + backward in one call, composed from the pieces above."""
        import torch
        logit = torch.zeros(B)
        lp = ctypes.c_void_p(logit.data_ptr())
        self.dctr_mlp_fwd(mref, x, ld_x, B, lp, stream)
        self.dctr_bce_head(p0, p1, lp, None, bias, y, B, y_pred, loss, g_logit, g_bias, stream)
        self.dctr_mlp_bwd(mref, x, ld_x, B, g_logit, 0, gx, ld_gx, ws, stream)
        self.calls = self.calls[:-3] + ["mlp_train_step"]
        return 0

    # ---- head ---------------------------------------------------------------------------------------------
    def dctr_bce_head(self, p0, p1, p2, p3, bias, y, B, y_pred, loss, g_logit, g_bias, stream):
        self.calls.append("bce_head")
        z = np.zeros(B, np.float32)
        for p in (p0, p1, p2, p3):
            if p is not None and p.value:
                z = z + _arr(p, (B,))
        if bias is not None and bias.value:
            z = z + _arr(bias, (1,))[0]
        p = sigmoid(z.astype(np.float64))
        t = _arr(y, (B,)).astype(np.float64)
        lp, l1p = np.maximum(np.log(np.maximum(p, 1e-300)), -100), np.maximum(np.log(np.maximum(1 - p, 1e-300)), -100)
        _arr(loss, (1,))[0] = np.sum((t - 1) * l1p - t * lp)
        _arr(y_pred, (B,))[...] = p
        gz = (p - t) / np.maximum((1 - p) * p, 1e-12) * ((1 - p) * p)
        _arr(g_logit, (B,))[...] = gz
        if g_bias is not None and g_bias.value:
            _arr(g_bias, (1,))[0] = gz.sum()
        return 0

    # ---- dense optimizer ----------------------------------------------------------------------------------
    def dctr_dense_opt(self, p, g, st, n, opt, lr, eps, stream):
        self.calls.append("dense_opt")
        P, G = _arr(p, (n,)), _arr(g, (n,))
        if opt == 1:
            S = _arr(st, (n,))
            S += G * G
            P -= lr * (G / (np.sqrt(S) + eps))
        else:
            P -= lr * G
        return 0

    def dctr_strerror(self, code):
        return b"mock error"
