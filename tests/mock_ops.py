"""Stand-ins for the interaction-layer entry points of libdctr_hip.so (CPU tests only; see tests/mock_lib.py).

Each forward restates the formula include/dctr.h documents for the entry point, with torch on the caller's host
buffers; each backward is torch.autograd of that forward -- no hand-derived gradients here, so these stand-ins are an
independent check of the argument marshalling in deepctr_torch/_hip/ops.py, not a second copy of the kernels' math."""
import itertools

import numpy as np
import torch

from mock_lib import _arr


def _t(ptr, rows, cols, ld=None):
    """torch view [rows, cols] of host memory (leading dimension ld), or None."""
    a = _arr(ptr, (rows, cols), ld)
    return None if a is None else torch.from_numpy(a)


def _v(ptr, n):
    a = _arr(ptr, (n,))
    return None if a is None else torch.from_numpy(a)


def _pairs(F):
    return list(itertools.combinations(range(F), 2))


def _grads(out, inputs, gout):
    """d (out . gout) / d inputs with zeros for inputs the output does not depend on."""
    live = [x for x in inputs if x is not None]
    gs = torch.autograd.grad(out, live, gout, allow_unused=True)
    it = iter(gs)
    return [None if x is None else (lambda g, x=x: torch.zeros_like(x) if g is None else g)(next(it)) for x in inputs]


def _with_grad(fn):
    """the stand-in backwards are called from autograd.Function.backward, where grad mode is off"""
    def wrapped(*a, **k):
        with torch.enable_grad():
            return fn(*a, **k)
    wrapped.__name__ = fn.__name__
    return wrapped


class OpsMixin(object):
    # ---- FM on an explicit tensor ---------------------------------------------------------------------------------
    @staticmethod
    def _fm(E):
        return 0.5 * (E.sum(1).pow(2) - E.pow(2).sum(1)).sum(1)

    def dctr_fm_fwd(self, E, ld_b, B, F, D, y, stream):
        self.calls.append("fm_fwd")
        _v(y, B).copy_(self._fm(_t(E, B, F * D, ld_b).reshape(B, F, D)))
        return 0

    @_with_grad
    def dctr_fm_bwd(self, E, ld_b, B, F, D, gy, gE, ld_gb, accumulate, stream):
        self.calls.append("fm_bwd")
        e = _t(E, B, F * D, ld_b).clone().requires_grad_(True)
        g, = _grads(self._fm(e.reshape(B, F, D)), [e], _v(gy, B))
        dst = _t(gE, B, F * D, ld_gb)
        dst.copy_(dst + g if accumulate else g)
        return 0

    # ---- BiInteractionPooling + NFM's DNN input -------------------------------------------------------------------
    @staticmethod
    def _bi(G, F, D, dense_off, n_dense):
        E = G[:, :F * D].reshape(-1, F, D)
        bi = 0.5 * (E.sum(1).pow(2) - E.pow(2).sum(1))
        return torch.cat([bi, G[:, dense_off:dense_off + n_dense]], 1) if n_dense else bi

    def dctr_bi_pooling_fwd(self, G, ld_g, B, F, D, dense_off, n_dense, out, ld_o, stream):
        self.calls.append("bi_pooling_fwd")
        width = max(F * D, dense_off + n_dense)
        _t(out, B, D + n_dense, ld_o).copy_(self._bi(_t(G, B, width, ld_g), F, D, dense_off, n_dense))
        return 0

    @_with_grad
    def dctr_bi_pooling_bwd(self, G, ld_g, B, F, D, dense_off, n_dense, gout, ld_go, gG, ld_gg, stream):
        self.calls.append("bi_pooling_bwd")
        width = max(F * D, dense_off + n_dense)
        g_in = _t(G, B, width, ld_g).clone().requires_grad_(True)
        g, = _grads(self._bi(g_in, F, D, dense_off, n_dense), [g_in], _t(gout, B, D + n_dense, ld_go))
        dst = _t(gG, B, width, ld_gg)
        dst[:, :F * D] = g[:, :F * D]
        if n_dense:
            dst[:, dense_off:dense_off + n_dense] = g[:, dense_off:dense_off + n_dense]
        return 0

    # ---- InnerProduct -----------------------------------------------------------------------------------------------
    @staticmethod
    def _inner(E, reduce):
        i, j = zip(*_pairs(E.shape[1])) if E.shape[1] > 1 else ((), ())
        p = E[:, list(i)] * E[:, list(j)]
        return p.sum(-1) if reduce else p.reshape(E.shape[0], -1)

    def dctr_inner_product_fwd(self, E, ld_e, B, F, D, reduce, out, ld_o, stream):
        self.calls.append("inner_product_fwd")
        P = F * (F - 1) // 2
        _t(out, B, P * (1 if reduce else D), ld_o).copy_(self._inner(_t(E, B, F * D, ld_e).reshape(B, F, D), reduce))
        return 0

    @_with_grad
    def dctr_inner_product_bwd(self, E, ld_e, B, F, D, reduce, gp, ld_g, gE, ld_ge, stream):
        self.calls.append("inner_product_bwd")
        P = F * (F - 1) // 2
        e = _t(E, B, F * D, ld_e).clone().requires_grad_(True)
        g, = _grads(self._inner(e.reshape(B, F, D), reduce), [e], _t(gp, B, P * (1 if reduce else D), ld_g))
        _t(gE, B, F * D, ld_ge).copy_(g)
        return 0

    # ---- autograd glue (csrc/head.hip: dctr_rows_join, dctr_relu_bwd_bias) ---------------------------------------------
    def dctr_rows_join(self, a, ld_a, c, ld_c, W, d, ld_d, n_d, out, ld_out, B, stream):
        self.calls.append("rows_join")
        o = _t(out, B, ld_out)
        res = torch.zeros_like(o)          # (out may be a itself: every word is read before it is written, like the kernel)
        if a:
            res[:, :W] += _t(a, B, W, ld_a)
        if c:
            res[:, :W] += _t(c, B, W, ld_c)
        if d and n_d > 0:
            res[:, W:W + n_d] = _t(d, B, n_d, ld_d)
        o.copy_(res)
        return 0

    def dctr_relu_bwd_bias_workspace_floats(self, B, N):
        return 16

    def dctr_relu_bwd_bias(self, g, ld_g, h, ld_h, B, N, g_out, ld_o, g_bias, ws, stream):
        self.calls.append("relu_bwd_bias")
        gv = _t(g, B, N, ld_g)
        v = gv * (_t(h, B, N, ld_h) > 0) if h else gv
        if g_out:
            _t(g_out, B, N, ld_o).copy_(v)
        _v(g_bias, N).copy_(v.sum(0))
        return 0

    # ---- SENET ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _senet(E, W1, W2):
        a1 = torch.relu(E.mean(-1) @ W1.t())
        a = torch.relu(a1 @ W2.t())
        return E * a[:, :, None], a, a1

    def dctr_senet_fwd(self, E, ld_e, B, F, D, W1, W2, R, V, a, a1, stream):
        self.calls.append("senet_fwd")
        v, av, a1v = self._senet(_t(E, B, F * D, ld_e).reshape(B, F, D), _t(W1, R, F), _t(W2, F, R))
        _t(V, B, F * D).copy_(v.reshape(B, -1))
        _t(a, B, F).copy_(av)
        _t(a1, B, R).copy_(a1v)
        return 0

    def dctr_senet_bwd_workspace_floats(self, B, F, R):
        return 16

    @_with_grad
    def dctr_senet_bwd(self, gV, E, ld_e, B, F, D, W1, W2, R, a, a1, gE, gW1, gW2, ws, stream):
        self.calls.append("senet_bwd")
        e = _t(E, B, F * D, ld_e).clone().requires_grad_(True)
        w1, w2 = _t(W1, R, F).clone().requires_grad_(True), _t(W2, F, R).clone().requires_grad_(True)
        v, _, _ = self._senet(e.reshape(B, F, D), w1, w2)
        ge, g1, g2 = _grads(v.reshape(B, -1), [e, w1, w2], _t(gV, B, F * D))
        _t(gE, B, F * D).copy_(ge)
        _t(gW1, R, F).copy_(g1)
        _t(gW2, F, R).copy_(g2)
        return 0

    # ---- Bilinear ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _bilinear(E, V, Wf, pair_w, dense, dense_off, width):
        B, F, D = E.shape
        i, j = zip(*_pairs(F))
        i, j = list(i), list(j)

        def one(X):                                    # (x_i W_w^T) * x_j
            W = Wf[pair_w]                             # [P, D, D]
            return (torch.einsum("bpd,ped->bpe", X[:, i], W) * X[:, j]).reshape(B, -1)
        parts = ([one(V)] if V is not None else []) + [one(E)]
        out = torch.cat(parts, 1)
        if dense is not None:
            pad = torch.zeros(B, width - out.shape[1])
            out = torch.cat([out, pad], 1)
            out = torch.cat([out[:, :dense_off], dense, out[:, dense_off + dense.shape[1]:]], 1)
        return out

    @staticmethod
    def _pair_w(sched, n_sched, P):
        s = _arr(sched, (n_sched * 4,), dtype=np.int32).reshape(n_sched, 4)
        pw = np.zeros(P, np.int64)
        for i, j, w, k in s:
            if i >= 0:
                pw[k] = w
        return torch.from_numpy(pw)

    def dctr_bilinear_fwd(self, E, ld_e, V, ld_v, Wf, sched, n_sched, P, F, D, B, out, ld_o, dense, ld_d, n_dense,
                          dense_off, stream):
        self.calls.append("bilinear_fwd")
        pw = self._pair_w(sched, n_sched, P)
        n_w = int(pw.max()) + 1
        e = _t(E, B, F * D, ld_e).reshape(B, F, D)
        v = _t(V, B, F * D, ld_v)
        v = v.reshape(B, F, D) if v is not None else None
        dn = _t(dense, B, n_dense, ld_d) if n_dense else None
        width = (2 if v is not None else 1) * P * D
        if dn is not None:
            width = max(width, dense_off + n_dense)
        res = self._bilinear(e, v, _t(Wf, n_w * D, D).reshape(n_w, D, D), pw, dn, dense_off, width)
        _t(out, B, width, ld_o).copy_(res)
        return 0

    def dctr_bilinear_bwd_workspace_floats(self, B, P, D):
        return 16

    @_with_grad
    def dctr_bilinear_bwd(self, E, ld_e, V, ld_v, Wf, sched, n_sched, slots, pair_w, n_w, P, F, D, B, gout, ld_g, gE, gV,
                          gW, ws, sched_k, n_sched_k, stream):
        self.calls.append("bilinear_bwd")
        pw = torch.from_numpy(_arr(pair_w, (P,), dtype=np.int32).astype(np.int64))
        e = _t(E, B, F * D, ld_e).clone().requires_grad_(True)
        v = _t(V, B, F * D, ld_v)
        v = v.clone().requires_grad_(True) if v is not None else None
        w = _t(Wf, n_w * D, D).clone().requires_grad_(True)
        width = (2 if v is not None else 1) * P * D
        res = self._bilinear(e.reshape(B, F, D), v.reshape(B, F, D) if v is not None else None, w.reshape(n_w, D, D), pw,
                             None, 0, width)
        ge, gv, gw = _grads(res, [e, v, w], _t(gout, B, width, ld_g))
        _t(gE, B, F * D).copy_(ge)
        if gv is not None:
            _t(gV, B, F * D).copy_(gv)
        _t(gW, n_w * D, D).copy_(gw)
        return 0

    # ---- CIN layer --------------------------------------------------------------------------------------------------
    @staticmethod
    def _cin(H, X0, W, bias, relu):
        """fp64 inside: a relu unit whose pre-activation is ~1e-8 must fall on the side the fp64 oracle (and, in the
        golden fixtures, the reference) puts it -- one such unit moves a weight gradient of xdeepfm_criteo by 4 %."""
        B, h, D = H.shape
        M = X0.shape[1]
        H, X0, W = H.double(), X0.double(), W.double()
        z = (H[:, :, None, :] * X0[:, None, :, :]).reshape(B, h * M, D)
        y = torch.einsum("oz,bzd->bod", W, z)
        if bias is not None:
            y = y + bias.double()[None, :, None]
        return (torch.relu(y) if relu else y).float()

    def dctr_cin_workspace_floats(self, h, M, O):
        return 16

    def dctr_cin_bwd_workspace_floats(self, B, h, M, D, O):
        return 16

    def dctr_cin_layer_fwd(self, H, ld_h, X0, ld_x0, W, bias, B, h, M, D, O, relu, A, ld_a, ws, stream):
        self.calls.append("cin_layer_fwd")
        y = self._cin(_t(H, B, h * D, ld_h).reshape(B, h, D), _t(X0, B, M * D, ld_x0).reshape(B, M, D), _t(W, O, h * M),
                      _v(bias, O), relu)
        _t(A, B, O * D, ld_a).copy_(y.reshape(B, -1))
        return 0

    @_with_grad
    def dctr_cin_layer_bwd(self, gA, A, ld_a, relu, H, ld_h, X0, ld_x0, W, B, h, M, D, O, gH, ld_gh, gX0, ld_gx,
                           accumulate_x0, gW, gbias, ws, stream):
        self.calls.append("cin_layer_bwd")
        hh = _t(H, B, h * D, ld_h).clone().requires_grad_(True)
        x0 = _t(X0, B, M * D, ld_x0).clone().requires_grad_(True)
        w = _t(W, O, h * M).clone().requires_grad_(True)
        bias = torch.zeros(O, requires_grad=True)
        g = _t(gA, B, O * D, ld_a)
        if relu:                       # the saved activation decides the mask (bias is not passed to the backward)
            g = g * (_t(A, B, O * D, ld_a) > 0)
        y = self._cin(hh.reshape(B, h, D), x0.reshape(B, M, D), w, bias, 0)
        gh, gx, gw, gb = _grads(y.reshape(B, -1), [hh, x0, w, bias], g)
        _t(gH, B, h * D, ld_gh).copy_(gh)
        dst = _t(gX0, B, M * D, ld_gx)
        dst.copy_(dst + gx if accumulate_x0 else gx)
        _t(gW, O, h * M).copy_(gw)
        if _v(gbias, O) is not None:
            _v(gbias, O).copy_(gb)
        return 0

    def dctr_cin_pool_fwd(self, A, B, O, D, pool_from, pooled, ld_pooled, stream):
        self.calls.append("cin_pool_fwd")
        a = _t(A, B, O * D).reshape(B, O, D)
        _t(pooled, B, O - pool_from, ld_pooled).copy_(a[:, pool_from:].double().sum(-1).float())
        return 0

    def dctr_cin_pool_bwd(self, g_hidden, g_pooled, ld_gp, w_head, A_relu, B, O, D, n_hidden, pool_from, gA, stream):
        """include/dctr.h: gA[b, o, :] = (o < n_hidden ? g_hidden[b, o, :] : 0) + (o >= pool_from ? gp(b, o - pool_from) : 0),
        gp = g_pooled[b, j], or g_pooled[b] * w_head[j] when the projection's backward is folded in; relu mask last."""
        self.calls.append("cin_pool_bwd")
        g = _t(gA, B, O * D).reshape(B, O, D)
        g.zero_()
        gh = _t(g_hidden, B, n_hidden * D) if n_hidden else None
        if gh is not None:
            g[:, :n_hidden] += gh.reshape(B, n_hidden, D)
        nd = O - pool_from
        if nd > 0 and _arr(g_pooled, (1,)) is not None:
            w = _v(w_head, nd)
            gp = _t(g_pooled, B, 1, ld_gp) * w[None, :] if w is not None else _t(g_pooled, B, nd, ld_gp)
            g[:, pool_from:] += gp[:, :, None]
        a = _t(A_relu, B, O * D)
        if a is not None:
            g *= (a.reshape(B, O, D) > 0).to(g.dtype)
        return 0

    def dctr_rows_dot(self, x, ld_x, w, B, N, out, stream):
        self.calls.append("rows_dot")
        _v(out, B).copy_((_t(x, B, N, ld_x).double() @ _v(w, N).double()).float())
        return 0

    def dctr_rows_tdot(self, x, ld_x, w, B, N, out, ws, stream):
        self.calls.append("rows_tdot")
        _v(out, N).copy_(torch.mm(_v(w, B).reshape(1, B), _t(x, B, N, ld_x)).reshape(-1))    # (the reference's own fp32 product)
        return 0

    # ---- CrossNet (vector) --------------------------------------------------------------------------------------------
    @staticmethod
    def _cross(X, K, Bv):
        x0 = xl = X
        for l in range(K.shape[0]):
            xl = x0 * (xl @ K[l])[:, None] + Bv[l] + xl
        return xl

    def dctr_crossnet_vec_fwd(self, X, ld_x, B, W, L, kernels, bias, Y, ld_y, stream):
        self.calls.append("crossnet_vec_fwd")
        _t(Y, B, W, ld_y).copy_(self._cross(_t(X, B, W, ld_x), _t(kernels, L, W), _t(bias, L, W)))
        return 0

    def dctr_crossnet_vec_bwd_workspace_floats(self, B, W, L):
        return 16

    @_with_grad
    def dctr_crossnet_vec_bwd(self, X, ld_x, B, W, L, kernels, bias, gY, ld_g, gX, ld_gx, g_kernels, g_bias, ws, stream):
        self.calls.append("crossnet_vec_bwd")
        x = _t(X, B, W, ld_x).clone().requires_grad_(True)
        k, b = _t(kernels, L, W).clone().requires_grad_(True), _t(bias, L, W).clone().requires_grad_(True)
        gx, gk, gb = _grads(self._cross(x, k, b), [x, k, b], _t(gY, B, W, ld_g))
        _t(gX, B, W, ld_gx).copy_(gx)
        _t(g_kernels, L, W).copy_(gk)
        _t(g_bias, L, W).copy_(gb)
        return 0

    # ---- CrossNet (matrix): x_{l+1} = x_0 (.) (x_l W_l^T + b_l) + x_l ------------------------------------------------
    def dctr_crossnet_mat_supported(self, W, n_layers):
        return 1 if (0 < W <= 512 and 0 < n_layers <= 8) else 0

    @staticmethod
    def _cross_mat_layers(mref):
        m = mref._obj
        return [m.layer[l] for l in range(m.n_layers)]

    def dctr_crossnet_mat_fwd(self, mref, x, ld_x, B, stream):
        self.calls.append("crossnet_mat_fwd")
        layers = self._cross_mat_layers(mref)
        W = layers[0].K
        x0 = xl = _t(x, B, W, ld_x).double()
        for e in layers:
            u = xl @ _t(e.W, W, W, e.ld_w).double().t() + _v(e.bias, W).double()
            _t(e.dh, B, W, e.ld_h).copy_(u)
            xl = x0 * u + xl
            _t(e.h, B, W, e.ld_h).copy_(xl)
        return 0

    def dctr_crossnet_mat_bwd_workspace_floats(self, mref, B):
        return 16

    @_with_grad
    def dctr_crossnet_mat_bwd(self, mref, x, ld_x, B, gY, ld_g, gx, ld_gx, ws, stream):
        self.calls.append("crossnet_mat_bwd")
        layers = self._cross_mat_layers(mref)
        W = layers[0].K
        x0 = _t(x, B, W, ld_x).double().clone().requires_grad_(True)
        Ws = [_t(e.W, W, W, e.ld_w).double().clone().requires_grad_(True) for e in layers]
        bs = [_v(e.bias, W).double().clone().requires_grad_(True) for e in layers]
        xl = x0
        for Wl, bl in zip(Ws, bs):
            xl = x0 * (xl @ Wl.t() + bl) + xl
        gs = _grads(xl, [x0] + Ws + bs, _t(gY, B, W, ld_g).double())
        _t(gx, B, W, ld_gx).copy_(gs[0])
        for l, e in enumerate(layers):
            full = _t(e.gW, W, e.ld_w, e.ld_w)
            full.zero_()
            full[:, :W].copy_(gs[1 + l])
            _v(e.gbias, W).copy_(gs[1 + len(layers) + l])
        return 0

    # ---- CrossNetMix on its packed weights (include/dctr.h): three dense layers per cross layer ------------------------
    def dctr_crossnet_mix_supported(self, W, n_cross, E, R):
        return 1 if (0 < W <= 512 and 0 < n_cross <= 4 and 0 < E <= 8 and E * R + E <= 512) else 0

    @staticmethod
    def _mix_chain(x0, packed, E, R):
        """packed: [(W1 [ER+E, W], W2 [ER, ER], W3 [W, ER], b [W])] per cross layer -> (x_L, per-layer saved tensors)"""
        ER = E * R
        xl, saved = x0, []
        for W1, W2, W3, b in packed:
            z = xl @ W1.t()
            v1, s = torch.tanh(z[:, :ER]), torch.softmax(z[:, ER:], dim=1)
            t = torch.tanh(v1 @ W2.t())
            ts = t * s.repeat_interleave(R, dim=1)
            u = ts @ W3.t() + b
            xl = x0 * u + xl
            saved.append((torch.cat([v1, s], 1), t, ts, u, xl))
        return xl, saved

    def dctr_crossnet_mix_fwd(self, mref, E, R, x, ld_x, B, stream):
        self.calls.append("crossnet_mix_fwd")
        m = mref._obj
        layers = [m.layer[l] for l in range(m.n_layers)]
        W, ER = layers[0].K, E * R
        packed = []
        for lc in range(m.n_layers // 3):
            e1, e2, e3 = layers[3 * lc:3 * lc + 3]
            packed.append((_t(e1.W, ER + E, W, e1.ld_w).double(), _t(e2.W, ER, ER, e2.ld_w).double(),
                           _t(e3.W, W, ER, e3.ld_w).double(), _v(e3.bias, W).double()))
        _, saved = self._mix_chain(_t(x, B, W, ld_x).double(), packed, E, R)
        for lc, (h1, t, ts, u, xn) in enumerate(saved):
            e1, e2, e3 = layers[3 * lc:3 * lc + 3]
            _t(e1.h, B, ER + E, e1.ld_h).copy_(h1)
            _t(e2.dh, B, ER, e2.ld_h).copy_(t)
            _t(e2.h, B, ER, e2.ld_h).copy_(ts)
            _t(e3.dh, B, W, e3.ld_h).copy_(u)
            _t(e3.h, B, W, e3.ld_h).copy_(xn)
        return 0

    def dctr_crossnet_mix_bwd_workspace_floats(self, mref, B):
        return 16

    @_with_grad
    def dctr_crossnet_mix_bwd(self, mref, E, R, x, ld_x, B, gY, ld_g, gx, ld_gx, ws, stream):
        self.calls.append("crossnet_mix_bwd")
        m = mref._obj
        layers = [m.layer[l] for l in range(m.n_layers)]
        W, ER = layers[0].K, E * R
        x0 = _t(x, B, W, ld_x).double().clone().requires_grad_(True)
        packed, leaves = [], []
        for lc in range(m.n_layers // 3):
            e1, e2, e3 = layers[3 * lc:3 * lc + 3]
            ws_ = [_t(e1.W, ER + E, W, e1.ld_w).double().clone().requires_grad_(True),
                   _t(e2.W, ER, ER, e2.ld_w).double().clone().requires_grad_(True),
                   _t(e3.W, W, ER, e3.ld_w).double().clone().requires_grad_(True),
                   _v(e3.bias, W).double().clone().requires_grad_(True)]
            packed.append(tuple(ws_))
            leaves += ws_
        out, _ = self._mix_chain(x0, packed, E, R)
        gs = _grads(out, [x0] + leaves, _t(gY, B, W, ld_g).double())
        _t(gx, B, W, ld_gx).copy_(gs[0])
        for lc in range(m.n_layers // 3):
            e1, e2, e3 = layers[3 * lc:3 * lc + 3]
            g1, g2, g3, gb = gs[1 + 4 * lc:5 + 4 * lc]
            for e, gw, rows, cols in ((e1, g1, ER + E, W), (e2, g2, ER, ER), (e3, g3, W, ER)):
                full = _t(e.gW, rows, e.ld_w, e.ld_w)
                full.zero_()
                full[:, :cols].copy_(gw)
            _v(e3.gbias, W).copy_(gb)
        return 0

    # ---- AFMLayer -----------------------------------------------------------------------------------------------------
    @staticmethod
    def _afm(E, W, bias, h, p):
        i, j = zip(*_pairs(E.shape[1]))
        bi = E[:, list(i)] * E[:, list(j)]                               # [B, P, D]
        s = torch.relu(bi @ W + bias) @ h                                # [B, P]
        a = torch.softmax(s, dim=1)
        return ((a[:, :, None] * bi).sum(1) * p).sum(1)

    def dctr_afm_bwd_workspace_floats(self, B, D, A):
        return 16

    def dctr_afm_fwd(self, E, ld_e, B, F, D, A, W, bias, h, p, y, stream):
        self.calls.append("afm_fwd")
        _v(y, B).copy_(self._afm(_t(E, B, F * D, ld_e).reshape(B, F, D), _t(W, D, A), _v(bias, A), _v(h, A), _v(p, D)))
        return 0

    @_with_grad
    def dctr_afm_bwd(self, E, ld_e, B, F, D, A, W, bias, h, p, gy, gE, ld_ge, gW, gbias, gh, gp, ws, stream):
        self.calls.append("afm_bwd")
        ins = [_t(E, B, F * D, ld_e), _t(W, D, A), _v(bias, A), _v(h, A), _v(p, D)]
        ins = [x.clone().requires_grad_(True) for x in ins]
        y = self._afm(ins[0].reshape(B, F, D), *ins[1:])
        ge, gw, gb, ghh, gpp = _grads(y, ins, _v(gy, B))
        _t(gE, B, F * D, ld_ge).copy_(ge)
        _t(gW, D, A).copy_(gw)
        _v(gbias, A).copy_(gb)
        _v(gh, A).copy_(ghh)
        _v(gp, D).copy_(gpp)
        return 0

    # ---- InteractingLayer ---------------------------------------------------------------------------------------------
    @staticmethod
    def _interact(E, H, scaling, Wq, Wk, Wv, Wr):
        B, F, D = E.shape
        A = D // H

        def heads(x):
            return x.reshape(B, F, H, A).permute(0, 2, 1, 3)             # [B, H, F, A]
        q, k, v = heads(E @ Wq), heads(E @ Wk), heads(E @ Wv)
        s = q @ k.transpose(-1, -2)
        if scaling:
            s = s / (A ** 0.5)
        o = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, F, D)
        if Wr is not None:
            o = o + E @ Wr
        return torch.relu(o).reshape(B, -1)

    def dctr_interacting_supported(self, F, D, H):
        return 1 if (D <= 32 and F <= 64 and D % H == 0) else 0

    def dctr_interacting_bwd_workspace_floats(self, B, D):
        return 16

    def dctr_interacting_fwd(self, E, ld_e, B, F, D, H, scaling, Wq, Wk, Wv, Wr, out, ld_o, stream):
        self.calls.append("interacting_fwd")
        _t(out, B, F * D, ld_o).copy_(self._interact(_t(E, B, F * D, ld_e).reshape(B, F, D), H, scaling, _t(Wq, D, D),
                                                     _t(Wk, D, D), _t(Wv, D, D), _t(Wr, D, D)))
        return 0

    @_with_grad
    def dctr_interacting_bwd(self, E, ld_e, B, F, D, H, scaling, Wq, Wk, Wv, Wr, gout, ld_g, gE, ld_ge, gWq, gWk, gWv,
                             gWr, ws, stream):
        self.calls.append("interacting_bwd")
        ins = [_t(E, B, F * D, ld_e), _t(Wq, D, D), _t(Wk, D, D), _t(Wv, D, D), _t(Wr, D, D)]
        ins = [None if x is None else x.clone().requires_grad_(True) for x in ins]
        out = self._interact(ins[0].reshape(B, F, D), H, scaling, *ins[1:])
        gs = _grads(out, ins, _t(gout, B, F * D, ld_g))
        _t(gE, B, F * D, ld_ge).copy_(gs[0])
        for ptr, g in zip((gWq, gWk, gWv, gWr), gs[1:]):
            if g is not None and _t(ptr, D, D) is not None:
                _t(ptr, D, D).copy_(g)
        return 0
