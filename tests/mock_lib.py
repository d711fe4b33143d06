"""A numpy stand-in for the gather / update / lazy-optimizer / tower / head / dense-optimizer entry points of
libdctr_hip.so, for CPU tests of the Python plumbing above the C-ABI (autograd Functions, EmbeddingPlan, LazyState,
DenseSlab, the fused train step, argument marshalling).  Test infrastructure only:
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
        ct = {np.int32: ctypes.c_int32, np.uint16: ctypes.c_uint16, np.uint8: ctypes.c_uint8}.get(dtype, ctypes.c_float)
        return np.ctypeslib.as_array((ct * n).from_address(addr))
    rows, cols = shape
    ld = cols if ld is None else int(ld)
    flat = np.ctypeslib.as_array((ctypes.c_float * (max(rows - 1, 0) * ld + cols)).from_address(addr))
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(ld * 4, 4))


def _tab(f):
    """the table of a field descriptor, honouring its row stride (dctr_field_t.ld; 0 = contiguous)"""
    return _arr(f.table, (f.vocab, f.dim), f.ld or f.dim)


def _st(f):
    return _arr(f.state, (f.vocab, f.dim), f.ld_state or f.dim)


class _Core(object):
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

    def dctr_mlp_train_wgrad_counters(self, mref, B):
        return 4

    @staticmethod
    def _dense_step(step, gptr, n):
        """dctr_dense_step_t applied to the n parameters behind the gradient at gptr (include/dctr.h)"""
        if step is None or not gptr:
            return
        s = step._obj
        addr = gptr.value if isinstance(gptr, ctypes.c_void_p) else int(gptr)
        k = addr - s.grad_base
        g = _arr(addr, (n,))
        p = _arr(s.param_base + k, (n,))
        if s.kind == 1:
            st = _arr(s.state_base + k, (n,))
            st += g * g
            p -= np.float32(s.lr) * (g / (np.sqrt(st) + np.float32(s.eps)))
        else:
            p -= np.float32(s.lr) * g

    def dctr_mlp_train_wgrad(self, mref, x, ld_x, B, g_logit, ws, loss, g_bias, step, stream):
        """the mock's train step has already produced the weight gradients; what is left is the in-kernel optimizer"""
        hp = getattr(self, "_head_partials", None)
        if hp is not None:
            self._head_partials = None
            _arr(loss, (1,))[0] = hp[0]
            if g_bias is not None and getattr(g_bias, "value", g_bias):
                _arr(g_bias, (1,))[0] = hp[1]
        if step is not None:
            self.calls.append("mlp_train_wgrad+step")
            m, layers = self._layers(mref)
            for e in layers:
                self._dense_step(step, e.gW, e.N * e.ld_w)
                if e.bias and e.gbias:
                    self._dense_step(step, e.gbias, e.N)
            if m.g_w_out:
                self._dense_step(step, m.g_w_out, layers[-1].N)
            if g_bias is not None and getattr(g_bias, "value", g_bias):
                self._dense_step(step, g_bias, 1)
        return 0

    def dctr_mlp_train_step(self, mref, x, ld_x, B, p0, p1, bias, y, y_pred, loss, g_logit, g_bias, gx, ld_gx, ws,
                            defer_wgrad, step, stream):
        """forward + head + backward in one call, composed from the pieces above."""
        import torch
        logit = torch.zeros(B)
        lp = ctypes.c_void_p(logit.data_ptr())
        self.dctr_mlp_fwd(mref, x, ld_x, B, lp, stream)
        self.dctr_bce_head(p0, p1, lp, None, bias, y, B, y_pred, loss, g_logit, g_bias, stream)
        self.dctr_mlp_bwd(mref, x, ld_x, B, g_logit, 0, gx, ld_gx, ws, stream)
        self.calls = self.calls[:-3] + ["mlp_train_step"]
        return 0

    def dctr_embed_tower_train_supported(self, pref, mref, B):
        c = pref._obj
        pooled = c.n_deep != c.n_deep_fixed or c.n_wide != c.n_wide_fixed
        if pooled and not c.ext:          # (pooled fields: their positions are listed in the ext block)
            return 0
        return int(c.n_deep >= 1 and c.n_deep_fixed >= 1 and c.n_wide <= 32 and
                   c.vec == 4 and c.emb_dim in (4, 8, 16, 32, 64) and
                   mref._obj.layer[0].K == c.n_deep * c.emb_dim + c.n_dense)

    def dctr_embed_tower_train_step(self, pref, X, ldx, mref, B, want_fm, bias, y, y_pred, g_logit, gx, ld_gx, out,
                                    ld_out, fm_s, ld_s, err, ws, stream):
        """dctr_embed_fwd + dctr_mlp_train_step(defer_wgrad) as one call (include/dctr.h), composed from the two."""
        import torch
        wide, fm, loss, gb = torch.zeros(B), torch.zeros(B), torch.zeros(1), torch.zeros(1)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.dctr_embed_fwd(pref, X, ldx, B, out, ld_out, P(wide), 1, P(fm) if want_fm else None, err, None, 0, None,
                            None, fm_s, ld_s, stream)
        self.dctr_mlp_train_step(mref, out, ld_out, B, P(wide), P(fm) if want_fm else None, bias, y, y_pred, P(loss),
                                 g_logit, P(gb), gx, ld_gx, ws, 1, None, stream)
        # (the device code leaves per-workgroup partial sums of the loss and of d loss / d bias in the workspace and the
        # weight-gradient call reduces them: the stand-in hands the finished values over the same way)
        self._head_partials = (float(loss[0]), float(gb[0]))
        self.calls = self.calls[:-2] + ["embed_tower_train_step"]
        return 0

    def dctr_stamp(self, dst, stream):
        return 0

    def dctr_step_signal(self, sync, signal, stream):
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

    def dctr_dense_opt_multi(self, items, n_items, opt, lr, eps, stream):
        for i in range(n_items):
            it = items[i]
            if it.n:
                if it.l2:
                    G = _arr(ctypes.c_void_p(it.g), (it.n,))
                    G += (np.float32(2.0) * np.float32(it.l2)) * _arr(ctypes.c_void_p(it.p), (it.n,))
                self.dctr_dense_opt(ctypes.c_void_p(it.p), ctypes.c_void_p(it.g),
                                    ctypes.c_void_p(it.state) if it.state else None, it.n, opt, lr, eps, stream)
        self.calls.append("dense_opt_multi")
        return 0

    def dctr_l2_value_multi(self, items, n_items, out, stream):
        tot = 0.0
        for i in range(n_items):
            it = items[i]
            if it.n and it.l2:
                P = _arr(ctypes.c_void_p(it.p), (it.n,)).astype(np.float64)
                tot += float(it.l2) * float((P * P).sum())
        _arr(out, (1,))[0] = tot
        return 0

    # ---- fused gather (include/dctr.h: dctr_embed_fwd), fixed-length fields only ----------------------------
    @staticmethod
    def _plan(pref):
        """dctr_plan_t -> (plan struct, deep fields, wide fields, dense cols, wide-dense cols)."""
        from deepctr_torch._hip import lib as L
        c = pref._obj
        deep = [(L.Field * c.n_deep).from_address(c.deep)[i] for i in range(c.n_deep)] if c.n_deep else []
        wide = [(L.Field * c.n_wide).from_address(c.wide)[i] for i in range(c.n_wide)] if c.n_wide else []
        dcols = list(_arr(c.dense_cols, (c.n_dense,), dtype=np.int32)) if c.n_dense else []
        wcols = list(_arr(c.wdense_cols, (c.n_wdense,), dtype=np.int32)) if c.n_wdense else []
        return c, deep, wide, dcols, wcols

    @staticmethod
    def _ext(pref):
        """dctr_plan_ext_t of a general plan -> (ext struct, slots, vunits), else None."""
        from deepctr_torch._hip import lib as L
        c = pref._obj if pref is not None else None
        if c is None or not c.ext:
            return None
        x = L.PlanExt.from_address(c.ext)
        return x, (L.USlot * x.n_vcols).from_address(x.slots), (L.VUnit * x.n_vunits).from_address(x.vunits)

    @staticmethod
    def _rows(X, f, err=None):
        ids = X[:, f.col].astype(np.int64)
        bad = (ids < 0) | (ids >= f.vocab)
        if err is not None and bad.any():
            err[0] = 1
        return np.where(bad, 0, ids)

    @staticmethod
    def _seq(X, f, err=None):
        """ids [B, len] (out-of-range -> 0) and the validity mask [B, len] of a field (inputs.py:146, sequence.py:56-59)."""
        ids = X[:, f.col:f.col + f.len].astype(np.int64)
        bad = (ids < 0) | (ids >= f.vocab)
        if err is not None and bad.any():
            err[0] = 1
        ids = np.where(bad, 0, ids)
        if f.pool == 0:
            return ids, np.ones_like(ids, dtype=bool)
        if f.len_col >= 0:
            return ids, np.arange(f.len)[None, :] < X[:, f.len_col].astype(np.int64)[:, None]
        return ids, ids != 0

    def _gather(self, X, f, err=None):
        """pooled embedding [B, dim] of one field (sequence.py:61-77), fp32 like the reference."""
        table = _tab(f)
        if f.pool == 0:                     # SparseFeat (a VarLen column of maxlen 1 still masks its padding id)
            return table[self._rows(X, f, err)]
        ids, mask = self._seq(X, f, err)
        rows = table[ids]                                          # [B, len, dim]
        m = mask[:, :, None].astype(np.float32)
        if f.pool == 3:
            return (rows - (np.float32(1) - m) * np.float32(1e9)).max(axis=1)
        tot = (rows * m).sum(axis=1, dtype=np.float32)
        if f.pool == 2:
            tot = tot / (mask.sum(axis=1, keepdims=True).astype(np.float32) + np.float32(1e-8))
        return tot

    def _scatter(self, X, f, G):
        """(row ids [n], gradient rows [n, dim]) of one field for the pooled gradient G [B, dim]."""
        if f.pool == 0:
            return self._rows(X, f), G
        ids, mask = self._seq(X, f)
        m = mask[:, :, None].astype(np.float32)
        if f.pool == 3:
            rows = _tab(f)[ids]
            arg = (rows - (np.float32(1) - m) * np.float32(1e9)).argmax(axis=1)      # [B, dim]
            g = np.zeros(rows.shape, np.float32)
            np.put_along_axis(g, arg[:, None, :], G[:, None, :], axis=1)
        else:
            g = G[:, None, :] * m
            if f.pool == 2:
                g = g / (mask.sum(axis=1).astype(np.float32)[:, None, None] + np.float32(1e-8))
        return ids.reshape(-1), g.reshape(-1, f.dim)

    def dctr_embed_update_supported(self, pref, max_vocab, B):
        return 1

    def dctr_embed_update_workspace_ints(self, pref, n_units, B):
        return 4

    def dctr_embed_segments(self, pref, units, n_units, max_vocab, ids_t, parts_t, B, ws, ws_n, stream):
        """(the pre-pass only re-orders work inside the device code: nothing to compute for the stand-in)"""
        self.calls.append("embed_segments")
        return 0

    def dctr_embed_update_partitions(self, pref, B):
        return max(1, (int(B) + 95) // 96)

    def dctr_embed_ids(self, pref, units, n_units, X, ldx, B, ids_t, parts_t, stream):
        """(parts_t -- the update kernel's partition tags -- is an optimisation detail of the device code: the
        stand-in's update does not read it, and leaves it unwritten)"""
        self.calls.append("embed_ids")
        ext = self._ext(pref)
        if ext is not None:
            # general units: ids_t / parts_t are [n_vcols, B]; a masked-out position carries the tag 0xFFFF (every other
            # tag is 0 here: the stand-in's update does not partition); mean pooling's divisors go to ext.den_t
            x, slots, _ = ext
            c = pref._obj
            Xv = _arr(X, (B, c.n_xcols), ldx)
            out = _arr(ids_t, (x.n_vcols * B,), dtype=np.int32).reshape(x.n_vcols, B)
            tags = _arr(parts_t, (x.n_vcols * B,), dtype=np.uint16)
            tags = tags.reshape(x.n_vcols, B) if tags is not None else None
            den = _arr(ctypes.c_void_p(x.den_t), (x.n_den * B,)).reshape(x.n_den, B) if (x.n_den and x.den_t) else None
            for ci in range(x.n_vcols):
                sl = slots[ci]
                rid = Xv[:, sl.col].astype(np.int32)
                out[ci] = rid
                valid = np.ones(B, bool)
                if sl.pool in (1, 2):
                    valid = (sl.t < Xv[:, sl.len_col].astype(np.int32)) if sl.len_col >= 0 else (rid != 0)
                if tags is not None:
                    tags[ci] = np.where(valid, 0, 0xFFFF).astype(np.uint16)
                if den is not None and sl.den >= 0 and sl.t == 0:
                    cnt = Xv[:, sl.len_col].astype(np.int32).astype(np.float32) if sl.len_col >= 0 else \
                        (Xv[:, sl.col:sl.col + sl.len].astype(np.int32) != 0).sum(axis=1).astype(np.float32)
                    den[sl.den] = cnt + np.float32(1e-8)
            return 0
        U = _arr(units, (n_units * 4,), dtype=np.int32).reshape(n_units, 4)
        ncol = int(U[:, 2].max()) + 1
        Xv = _arr(X, (B, ncol), ldx)
        out = _arr(ids_t, (n_units * B,), dtype=np.int32).reshape(n_units, B)
        for u in range(n_units):
            out[u] = Xv[:, U[u, 2]].astype(np.int32)
        return 0

    def dctr_embed_fwd(self, pref, X, ldx, B, out, ld_out, wide, ld_wide, fm, err, units, n_units, ids_t, parts_t,
                       fm_s, ld_s, stream):
        self.calls.append("embed_fwd")
        c, deep, widef, dcols, wcols = self._plan(pref)
        Xv = _arr(X, (B, c.n_xcols), ldx)
        errv = _arr(err, (1,), dtype=np.int32)
        S = np.zeros((B, max(c.emb_dim, 1)), np.float32)
        Q = np.zeros_like(S)
        if out is not None and _arr(out, (1,)) is not None:
            O = _arr(out, (B, ld_out), ld_out)
            O[...] = 0
            for f in deep:
                e = self._gather(Xv, f, errv)
                O[:, f.out_off:f.out_off + f.dim] = e
                if c.emb_dim > 0:
                    S += e
                    Q += e * e
            for j, col in enumerate(dcols):
                O[:, c.dense_off + j] = Xv[:, col]
        if _arr(wide, (1,)) is not None:
            w = np.zeros(B, np.float32)
            for f in widef:
                w += self._gather(Xv, f, errv)[:, 0]
            if wcols:
                ww = _arr(c.wdense_w, (len(wcols),))
                for j, col in enumerate(wcols):
                    w += Xv[:, col] * ww[j]
            _arr(wide, ((B - 1) * ld_wide + 1,))[::ld_wide] = w
        if _arr(fm, (1,)) is not None:
            _arr(fm, (B,))[...] = 0.5 * np.sum(S * S - Q, axis=1)
        if _arr(fm_s, (1,)) is not None:
            Sv = _arr(fm_s, (B, ld_s), ld_s)
            Sv[...] = 0
            Sv[:, :c.emb_dim] = S[:, :c.emb_dim]
        if _arr(ids_t, (1,)) is not None:
            self.dctr_embed_ids(pref, units, n_units, X, ldx, B, ids_t, parts_t, stream)
            self.calls.pop()
        ext = self._ext(pref)
        if ext is not None and ext[0].amax and ext[0].ld_amax:
            x = ext[0]
            A = _arr(ctypes.c_void_p(x.amax), (B * x.ld_amax,), dtype=np.uint8).reshape(B, x.ld_amax)
            offd = _arr(ctypes.c_void_p(x.am_deep_off), (max(1, c.n_deep),), dtype=np.int32)
            offw = _arr(ctypes.c_void_p(x.am_wide_off), (max(1, c.n_wide),), dtype=np.int32)
            for fs, offs in ((deep, offd), (widef, offw)):
                for i, f in enumerate(fs):
                    if f.pool == 3 and offs[i] >= 0:
                        ids, mask = self._seq(Xv, f)
                        m = mask[:, :, None].astype(np.float32)
                        arg = (_tab(f)[ids] - (np.float32(1) - m) * np.float32(1e9)).argmax(axis=1)      # first maximum
                        A[:, offs[i]:offs[i] + f.dim] = arg.astype(np.uint8)
        return 0

    # ---- general backward (dctr_embed_bwd) + consume pass (dctr_embed_apply): pooled fields, shared tables -----
    def dctr_embed_bwd(self, pref, X, ldx, B, g_out, ld_g, out, ld_out, g_fm, g_wide, mode, lr, stream):
        self.calls.append("embed_bwd:%d" % mode)
        c, deep, widef, dcols, wcols = self._plan(pref)
        Xv = _arr(X, (B, c.n_xcols), ldx)
        gO = _arr(g_out, (B, ld_g), ld_g) if ld_g else None
        gF, gW = _arr(g_fm, (B,)), _arr(g_wide, (B,))
        O = _arr(out, (B, ld_out), ld_out) if gF is not None else None
        S = sum(O[:, f.out_off:f.out_off + f.dim] for f in deep) if gF is not None else None
        work = []
        for f in deep:
            G = np.zeros((B, f.dim), np.float32)
            if gO is not None:
                G += gO[:, f.out_off:f.out_off + f.dim]
            if gF is not None:
                G += gF[:, None] * (S - O[:, f.out_off:f.out_off + f.dim])
            work.append((f, G))
        if gW is not None:
            work += [(f, gW.reshape(B, 1).astype(np.float32)) for f in widef]
        scat = [(f,) + self._scatter(Xv, f, G) for f, G in work]       # max pooling re-reads the tables: gather first
        for f, rows, g in scat:
            dst = _arr(f.gacc, (f.vocab, f.dim)) if mode == 0 else _tab(f)
            np.add.at(dst, rows, g if mode == 0 else -np.float32(lr) * g)
        return 0

    def dctr_embed_apply(self, pref, X, ldx, B, opt, lr, eps, stream):
        self.calls.append("embed_apply:%d" % opt)
        c, deep, widef, dcols, wcols = self._plan(pref)
        Xv = _arr(X, (B, c.n_xcols), ldx)
        for f in deep + widef:
            rows = np.unique(self._seq(Xv, f)[0])
            gacc, table = _arr(f.gacc, (f.vocab, f.dim)), _tab(f)
            G = gacc[rows].copy()
            gacc[rows] = 0
            if opt == 1:
                st = _st(f)
                s2 = st[rows] + G * G
                table[rows] = np.where(G != 0, table[rows] - np.float32(lr) * (G / (np.sqrt(s2) + np.float32(eps))),
                                       table[rows])
                st[rows] = s2
            else:
                table[rows] -= np.float32(lr) * G
        return 0

    # ---- deterministic fused backward + optimizer (dctr_embed_update) ---------------------------------------
    def dctr_embed_update_lazy(self, *a):
        return -2          # (DCTR_ENOSUP: the stand-in keeps the two-pass route)

    def dctr_embed_update(self, pref, units, n_units, max_vocab, ids_t, parts_t, B, g_out, ld_g, out, ld_out, fm_s, ld_s, g_fm,
                          g_wide, ld_gw, opt, lr, eps, X, ld_x, g_wdense, wd_step, ws, ws_n, presorted, stream):
        self.calls.append("embed_update:%d" % opt)
        c, deep, widef, dcols, wcols = self._plan(pref)
        gO = _arr(g_out, (B, ld_g), ld_g) if ld_g else None
        gF = _arr(g_fm, (B,))
        gW = _arr(g_wide, ((B - 1) * ld_gw + 1,))[::ld_gw] if _arr(g_wide, (1,)) is not None else None
        ext = self._ext(pref)
        if ext is not None:
            self._update_general(ext, c, deep, widef, ids_t, parts_t, B, gO, gF, gW, out, ld_out, fm_s, ld_s, opt, lr, eps)
            if _arr(g_wdense, (1,)) is not None and gW is not None and wcols:
                Xv = _arr(X, (B, c.n_xcols), ld_x)
                _arr(g_wdense, (len(wcols),))[...] = [np.dot(gW.astype(np.float64), Xv[:, col]) for col in wcols]
                self._dense_step(wd_step, g_wdense, len(wcols))
            return 0
        U = _arr(units, (n_units * 4,), dtype=np.int32).reshape(n_units, 4)
        ids = _arr(ids_t, (n_units * B,), dtype=np.int32).reshape(n_units, B)

        def scatter(f, rows, G):
            """sum duplicates in sample order, then one read-modify-write per touched row."""
            table = _tab(f)
            uniq, inv = np.unique(rows, return_inverse=True)
            acc = np.zeros((len(uniq), f.dim), np.float32)
            for b in range(B):
                acc[inv[b]] += G[b]
            if opt == 0:
                table[uniq] -= np.float32(lr) * acc
            elif opt == 1:
                st = _st(f)
                st[uniq] += acc * acc
                table[uniq] -= np.float32(lr) * (acc / (np.sqrt(st[uniq]) + np.float32(eps)))
            else:
                _arr(f.gacc, (f.vocab, f.dim))[uniq] += acc

        for u in range(n_units):
            di, wi = int(U[u, 0]), int(U[u, 1])
            if di >= 0:
                f = deep[di]
                rows = np.where((ids[u] < 0) | (ids[u] >= f.vocab), 0, ids[u]).astype(np.int64)
                G = np.zeros((B, f.dim), np.float32)
                if gO is not None:
                    G += gO[:, f.out_off:f.out_off + f.dim]
                if gF is not None:
                    e = _arr(out, (B, ld_out), ld_out)[:, f.out_off:f.out_off + f.dim]
                    G += gF[:, None] * (_arr(fm_s, (B, ld_s), ld_s)[:, :f.dim] - e)
                scatter(f, rows, G)
            if wi >= 0 and gW is not None:
                f = widef[wi]
                rows = np.where((ids[u] < 0) | (ids[u] >= f.vocab), 0, ids[u]).astype(np.int64)
                scatter(f, rows, gW.reshape(B, 1).astype(np.float32))
        if _arr(g_wdense, (1,)) is not None and gW is not None and wcols:
            Xv = _arr(X, (B, c.n_xcols), ld_x)
            _arr(g_wdense, (len(wcols),))[...] = [np.dot(gW.astype(np.float64), Xv[:, col]) for col in wcols]
            self._dense_step(wd_step, g_wdense, len(wcols))
        return 0

    def _update_general(self, ext, c, deep, widef, ids_t, parts_t, B, gO, gF, gW, out, ld_out, fm_s, ld_s, opt, lr, eps):
        """dctr_embed_update on GENERAL units (include/dctr.h, dctr_plan_ext_t): every valid (slot, sample) entry of a unit
        contributes its field's gradient slice times the pooling weight; duplicates of a row are summed in (id, slot, sample)
        order, each touched row is read-modify-written once."""
        x, slots, vunits = ext
        ids = _arr(ids_t, (x.n_vcols * B,), dtype=np.int32).reshape(x.n_vcols, B)
        tags = _arr(parts_t, (x.n_vcols * B,), dtype=np.uint16).reshape(x.n_vcols, B)
        O = _arr(out, (B, ld_out), ld_out) if gF is not None else None
        S = _arr(fm_s, (B, ld_s), ld_s) if gF is not None else None
        den = _arr(ctypes.c_void_p(x.den_t), (x.n_den * B,)).reshape(x.n_den, B) if (x.n_den and x.den_t) else None
        am = _arr(ctypes.c_void_p(x.amax), (B * x.ld_amax,), dtype=np.uint8).reshape(B, x.ld_amax) if (x.ld_amax and x.amax) \
            else None

        def apply(f, rows, G, fold_e=None):
            table = _tab(f)
            order = np.lexsort((np.arange(len(rows)), rows))
            rows, G = rows[order], G[order]
            uniq, start = np.unique(rows, return_index=True)
            acc = np.zeros((len(uniq), f.dim), np.float32)
            ends = list(start[1:]) + [len(rows)]
            for i, (a, b) in enumerate(zip(start, ends)):
                for r in range(b - 1, a - 1, -1):           # (the kernels walk a segment backwards)
                    acc[i] += G[r]
            if fold_e is not None:
                gfs = fold_e[order]
                for i, (a, b) in enumerate(zip(start, ends)):
                    tot = np.float32(0)
                    for r in range(b - 1, a - 1, -1):
                        tot += gfs[r]
                    acc[i] -= tot * table[uniq[i]]
            if opt == 0:
                table[uniq] -= np.float32(lr) * acc
            elif opt == 1:
                st = _st(f)
                st[uniq] += acc * acc
                table[uniq] -= np.float32(lr) * (acc / (np.sqrt(st[uniq]) + np.float32(eps)))
            else:
                _arr(f.gacc, (f.vocab, f.dim))[uniq] += acc

        work = []
        for vu in vunits:
            if vu.j != 0:
                continue
            fd = deep[vu.di] if vu.di >= 0 else None
            fw = widef[vu.wi] if vu.wi >= 0 else None
            vocab = (fd or fw).vocab
            rows_all, Gd_all, Gf_all, Gw_all = [], [], [], []
            for ci in range(vu.c0, vu.c0 + vu.n_slots):
                sl = slots[ci]
                valid = tags[ci] != 0xFFFF
                rid = ids[ci].astype(np.int64)
                rows = np.where((rid < 0) | (rid >= vocab), 0, rid)
                Gd = np.zeros((B, fd.dim if fd is not None else 1), np.float32)
                gfe = np.zeros(B, np.float32)
                if fd is not None and sl.goff >= 0 and (gO is not None or gF is not None):
                    D = fd.dim
                    if gO is not None:
                        Gd += gO[:, sl.goff:sl.goff + D]
                    if gF is not None:
                        if sl.pool == 0:
                            Gd += gF[:, None] * S[:, :D]
                            gfe = gF.astype(np.float32).copy()
                        else:
                            Gd += gF[:, None] * (S[:, :D] - O[:, sl.goff:sl.goff + D])
                    if sl.pool == 2:
                        Gd = Gd / den[sl.den][:, None]
                    elif sl.pool == 3:
                        Gd = np.where(am[:, sl.am_deep:sl.am_deep + D] == sl.t, Gd, np.float32(0))
                Gw = np.zeros((B, 1), np.float32)
                if fw is not None and sl.wide and gW is not None:
                    Gw = gW.reshape(B, 1).astype(np.float32)
                    if sl.pool == 2:
                        Gw = Gw / den[sl.den][:, None]
                    elif sl.pool == 3:
                        Gw = np.where(am[:, sl.am_wide:sl.am_wide + 1] == sl.t, Gw, np.float32(0))
                rows_all.append(rows[valid]); Gd_all.append(Gd[valid]); Gf_all.append(gfe[valid]); Gw_all.append(Gw[valid])
            rows = np.concatenate(rows_all)
            if not len(rows):
                continue
            work.append((fd, fw, rows, np.concatenate(Gd_all), np.concatenate(Gf_all), np.concatenate(Gw_all)))
        for fd, fw, rows, Gd, Gf, Gw in work:
            if fd is not None and (gO is not None or gF is not None):
                apply(fd, rows, Gd, Gf if gF is not None else None)
            if fw is not None and gW is not None:
                apply(fw, rows, Gw)

    # ---- exact lazy regularised / Adam update (csrc/lazy.hip) -------------------------------------------------
    @staticmethod
    def _adam_scalars(o, T):
        if o.adam_ss and o.adam_bc:       # the host's tables (dctr_lazy_opt_t): entry T - 1, the last entry past the end
            ss = _arr(ctypes.c_void_p(o.adam_ss), (o.n_ss,))
            bc = _arr(ctypes.c_void_p(o.adam_bc), (o.n_bc,))
            return ss[min(T, o.n_ss) - 1], bc[min(T, o.n_bc) - 1]
        return (np.float32(float(o.lr) / (1.0 - float(o.beta1) ** T)), np.float32(np.sqrt(1.0 - float(o.beta2) ** T)))

    @staticmethod
    def _opt_step(o, g, w, a, b, ss, bc):
        """torch.optim's single-tensor step in fp32 on arrays (in place); a / b may be None."""
        f = np.float32
        if o.kind == 2:
            a += (g - a) * (f(1) - f(o.beta1))
            b *= f(o.beta2)
            b += (f(1) - f(o.beta2)) * g * g
            w -= ss * (a / (np.sqrt(b) / bc + f(o.eps)))
        elif o.kind == 1:
            a += g * g
            w -= f(o.lr) * (g / (np.sqrt(a) + f(o.eps)))
        elif o.kind == 3:                       # RMSprop: beta2 = alpha, beta1 = 1 - alpha
            a *= f(o.beta2)
            a += (f(o.beta1) * g) * g
            w -= f(o.lr) * (g / (np.sqrt(a) + f(o.eps)))
        else:
            w -= f(o.lr) * g

    def _lazy_pass(self, mode, units, n_units, ids_t, n, step, optref, sweep_k=0):
        from deepctr_torch._hip import lib as L
        o = optref._obj
        t = int(_arr(step, (1,), dtype=np.int32)[0])
        target = t + 1 if mode == 1 else t
        arr = (L.LazyUnit * n_units).from_address(units.value)
        ids = _arr(ids_t, (n_units * n,), dtype=np.int32).reshape(n_units, n) if mode != 2 else None
        for u in range(n_units):
            un = arr[u]
            stamp = _arr(un.stamp, (un.vocab,), dtype=np.int32)
            if mode == 2 and sweep_k > 0:       # dctr_lazy_sweep: the (t mod K)-th of K windows of this table's rows
                wlen = (un.vocab + sweep_k - 1) // sweep_k
                rows = np.arange((t % sweep_k) * wlen, min((t % sweep_k + 1) * wlen, un.vocab))
            elif mode == 2:
                rows = np.arange(min(n, un.vocab))
            else:
                rows = np.unique(np.where((ids[u] < 0) | (ids[u] >= un.vocab), 0, ids[u]).astype(np.int64))
            rows = rows[stamp[rows] < target]
            if not len(rows):
                continue
            prev = stamp[rows].copy()
            stamp[rows] = target
            for tab, s1, s2, gp, dim, lam, ld_w, ld_a in (
                    (un.deep, un.deep_s1, un.deep_s2, un.deep_g, un.dim, un.l2_deep, un.ld_deep, un.ld_deep_s1),
                    (un.wide, un.wide_s1, un.wide_s2, un.wide_g, 1, un.l2_wide, un.ld_wide, un.ld_wide_s1)):
                if not tab:
                    continue
                W = _arr(tab, (un.vocab, dim), ld_w or dim)
                A = _arr(s1, (un.vocab, dim), ld_a or dim) if s1 else None
                Bv = _arr(s2, (un.vocab, dim)) if s2 else None
                lam2 = np.float32(2) * np.float32(lam)
                w = W[rows].copy()
                a = A[rows].copy() if A is not None else np.zeros_like(w)
                b = Bv[rows].copy() if Bv is not None else np.zeros_like(w)
                if o.kind in (2, 3) or lam2 != 0:               # replay the missed steps prev+1 .. t with g = 2*lambda*w
                    for T in range(int(prev.min()) + 1, t + 1):
                        m = prev < T
                        ss, bc = self._adam_scalars(o, T) if o.kind == 2 else (0, 1)
                        wm, am, bm = w[m], a[m], b[m]
                        self._opt_step(o, lam2 * wm, wm, am, bm, ss, bc)
                        w[m], a[m], b[m] = wm, am, bm
                if mode == 1:
                    Gv = _arr(gp, (un.vocab, dim))
                    g = Gv[rows] + lam2 * w
                    Gv[rows] = 0
                    ss, bc = self._adam_scalars(o, t + 1) if o.kind == 2 else (0, 1)
                    self._opt_step(o, g, w, a, b, ss, bc)
                W[rows] = w
                if A is not None:
                    A[rows] = a
                if Bv is not None:
                    Bv[rows] = b
        return 0

    def dctr_lazy_catchup(self, units, n_units, ids_t, B, step, opt, vec, max_dim, order_ws, stream):
        self.calls.append("lazy_catchup")
        return self._lazy_pass(0, units, n_units, ids_t, B, step, opt)

    def dctr_lazy_apply(self, units, n_units, ids_t, B, step, opt, vec, max_dim, stream):
        self.calls.append("lazy_apply")
        return self._lazy_pass(1, units, n_units, ids_t, B, step, opt)

    def dctr_lazy_flush(self, units, n_units, max_vocab, step, opt, vec, max_dim, stream):
        self.calls.append("lazy_flush")
        return self._lazy_pass(2, units, n_units, None, max_vocab, step, opt)

    def dctr_lazy_sweep(self, units, n_units, max_vocab, K, step, opt, vec, max_dim, stream):
        self.calls.append("lazy_sweep")
        return self._lazy_pass(2, units, n_units, None, max_vocab, step, opt, sweep_k=int(K))

    def dctr_lazy_step_inc(self, step, stream):
        _arr(step, (1,), dtype=np.int32)[0] += 1
        return 0

    def dctr_dense_opt_reg(self, p, g, s1, s2, lam, n, optref, step, stream):
        self.calls.append("dense_opt_reg")
        o = optref._obj
        T = int(_arr(step, (1,), dtype=np.int32)[0]) + 1
        P, G = _arr(p, (n,)), _arr(g, (n,))
        A = _arr(s1, (n,)) if _arr(s1, (1,)) is not None else np.zeros(n, np.float32)
        Bv = _arr(s2, (n,)) if _arr(s2, (1,)) is not None else np.zeros(n, np.float32)
        gt = G + (np.float32(2) * _arr(lam, (n,)) * P if _arr(lam, (1,)) is not None else np.float32(0))
        ss, bc = self._adam_scalars(o, T) if o.kind == 2 else (0, 1)
        self._opt_step(o, gt, P, A, Bv, ss, bc)
        return 0

    def dctr_strerror(self, code):
        return b"mock error"


def _make():
    from mock_ops import OpsMixin

    class MockLib(OpsMixin, _Core):
        """every entry point the Python stack calls: gather / update / lazy / tower / head (here) + interaction layers
        (tests/mock_ops.py)"""
    return MockLib


def MockLib():
    return _make()()
