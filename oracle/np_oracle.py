"""CPU oracle: a numpy restatement of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deepctr-torch_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may, and only as the checker.

Every function restates one piece of shenweichen/DeepCTR-Torch v0.2.9 (file:line under
``deepctr_torch/`` cited per function) forward AND backward (what autograd would compute), in plain
numpy.  The oracle is PINNED: ``tests/test_oracle_golden.py`` checks it against fixtures under
``tests/golden/`` that ``oracle/make_golden.py`` produced by executing the real reference
(``/root/reference``, torch-CPU fp32) in the build container -- the reference ships no golden vectors of
its own, so those fixtures are the pin.

Parameters are addressed by the reference's ``state_dict`` keys.  A model is described by a ``spec``:
    {"model": "DeepFM" | "xDeepFM" | "FiBiNET" | "DCN" | "PNN" | "NFM" | "AFM" | "WDL" | "AutoInt" | "DCNMix",
     "linear_columns": [col...], "dnn_columns": [col...], "kwargs": {...}}
with ``col`` = {"kind": "sparse"|"varlen"|"dense", "name", "vocab", "dim", "embedding_name",
                "maxlen", "combiner", "length_name", "dimension"}.
"""
import itertools
from collections import OrderedDict

import numpy as np


# --------------------------------------------------------------------------------------------------
# schema
# --------------------------------------------------------------------------------------------------
def build_input_features(columns):
    """name -> (start, end) columns of X.  inputs.py:99-123."""
    feats, start = OrderedDict(), 0
    for c in columns:
        if c["name"] in feats:
            continue
        if c["kind"] == "sparse":
            feats[c["name"]] = (start, start + 1)
            start += 1
        elif c["kind"] == "dense":
            feats[c["name"]] = (start, start + c["dimension"])
            start += c["dimension"]
        elif c["kind"] == "varlen":
            feats[c["name"]] = (start, start + c["maxlen"])
            start += c["maxlen"]
            if c.get("length_name") is not None and c["length_name"] not in feats:
                feats[c["length_name"]] = (start, start + 1)
                start += 1
        else:
            raise TypeError(c["kind"])
    return feats


def _split(columns):
    return ([c for c in columns if c["kind"] == "sparse"], [c for c in columns if c["kind"] == "varlen"],
            [c for c in columns if c["kind"] == "dense"])


def _ids(X, lo, hi):
    """``X[:, lo:hi].long()``: truncation toward zero.  basemodel.py:369, inputs.py:225."""
    return np.trunc(X[:, lo:hi]).astype(np.int64)


# --------------------------------------------------------------------------------------------------
# embedding lookup + VarLen pooling.  basemodel.py:354-380, inputs.py:141-155,213-227, sequence.py:49-77
# --------------------------------------------------------------------------------------------------
def _pool_mask(c, X, fi):
    lo, hi = fi[c["name"]]
    ids = _ids(X, lo, hi)                                        # [B, T]
    if c.get("length_name") is None:
        mask = (ids != 0)                                        # inputs.py:146
        length = mask.sum(axis=1, keepdims=True).astype(X.dtype)  # sequence.py:53
    else:
        llo, lhi = fi[c["length_name"]]
        seq_len = _ids(X, llo, lhi)                              # [B, 1]
        mask = np.arange(hi - lo)[None, :] < seq_len             # sequence.py:38-47
        length = seq_len.astype(X.dtype)
    return ids, mask, length


def lookup_forward(X, columns, fi, tables, prefix):
    """list of pooled [B, D] embeddings: sparse columns first, then VarLen (basemodel.py:380)."""
    sparse, varlen, _ = _split(columns)
    embs, cache = [], []
    dt = X.dtype
    for c in sparse:
        W = tables[prefix + c["embedding_name"] + ".weight"]
        ids = _ids(X, *fi[c["name"]])[:, 0]
        embs.append(W[ids].astype(dt))
        cache.append(("sparse", c, ids, None, None))
    for c in varlen:
        W = tables[prefix + c["embedding_name"] + ".weight"]
        ids, mask, length = _pool_mask(c, X, fi)
        seq = W[ids].astype(dt)                                  # [B, T, D]
        m = mask[:, :, None].astype(dt)
        if c["combiner"] == "max":                               # sequence.py:65-68
            hist = seq - (1 - m) * dt.type(1e9)
            arg = hist.argmax(axis=1)                            # first max wins, like torch CPU
            pooled = np.take_along_axis(hist, arg[:, None, :], axis=1)[:, 0, :]
            cache.append(("max", c, ids, arg, None))
        else:
            pooled = (seq * m).sum(axis=1)
            if c["combiner"] == "mean":                          # sequence.py:72-74
                pooled = pooled / (length + dt.type(1e-8))
            cache.append((c["combiner"], c, ids, mask, length))
        embs.append(pooled.astype(dt))
    return embs, cache


def lookup_backward(g_embs, cache, tables, prefix, grads):
    """Scatter the pooled-embedding gradients into dense [V, D] table gradients (duplicates add):
    aten::embedding_dense_backward + the pooling backward."""
    for g, (kind, c, ids, aux, length) in zip(g_embs, cache):
        key = prefix + c["embedding_name"] + ".weight"
        G = grads.setdefault(key, np.zeros_like(tables[key], dtype=g.dtype))
        if kind == "sparse":
            np.add.at(G, ids, g)
        elif kind == "max":
            B, D = g.shape
            rows = np.take_along_axis(ids, aux, axis=1)          # [B, D] row id that won per element
            np.add.at(G, (rows.reshape(-1), np.tile(np.arange(D), B)), g.reshape(-1))
        else:
            gs = g / (length + g.dtype.type(1e-8)) if kind == "mean" else g
            B, T = ids.shape
            for t in range(T):
                sel = aux[:, t]
                np.add.at(G, ids[sel, t], gs[sel])


# --------------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------------
def fm_forward(E):
    """interaction.py:26-34.  E [B, F, D] -> [B, 1]."""
    s = E.sum(axis=1)
    return 0.5 * ((s * s) - (E * E).sum(axis=1)).sum(axis=1, keepdims=True)


def fm_backward(E, gy):
    return gy[:, :, None] * (E.sum(axis=1, keepdims=True) - E)


def dnn_forward(x, P, prefix, n_layers):
    """core.py:120-134 with relu, no BN, no dropout."""
    acts = [x]
    for i in range(n_layers):
        W, b = P[prefix + "linears.%d.weight" % i], P[prefix + "linears.%d.bias" % i]
        x = np.maximum(x @ W.T + b, 0)
        acts.append(x)
    return x, acts


def relu_bias_backward(g, h):
    """What autograd does behind ``relu(x W^T + b)`` (core.py:120-134): aten::threshold_backward -- the gradient passes where
    the activation is positive -- and the bias gradient, the sum over the batch.  ``h`` None: no activation."""
    gz = g * (h > 0) if h is not None else g
    return gz, gz.sum(axis=0)


def dnn_backward(g, acts, P, prefix, n_layers, grads):
    for i in reversed(range(n_layers)):
        W = P[prefix + "linears.%d.weight" % i]
        g, gb = relu_bias_backward(g, acts[i + 1])
        grads[prefix + "linears.%d.weight" % i] = g.T @ acts[i]
        grads[prefix + "linears.%d.bias" % i] = gb
        g = g @ W
    return g


def optimizer_step(kind, p, g, state, lr, eps=1e-10, l2=0.0):
    """One step of the reference's optimizers on one tensor (basemodel.py:447-461: torch.optim.SGD(lr) / Adagrad(lr), default
    arguments), with an optional L2 term lambda * sum(p^2) of get_regularization_loss (basemodel.py:412-428) entering through
    its gradient 2 lambda p.  Returns (new p, new state); ``state`` is Adagrad's ``sum`` (None for SGD)."""
    if l2:
        g = g + 2.0 * l2 * p
    if kind == "sgd":
        return p - lr * g, state
    if kind == "adagrad":
        s = (state if state is not None else np.zeros_like(p)) + g * g
        return p - lr * g / (np.sqrt(s) + eps), s
    raise ValueError(kind)


def inner_product_forward(E):
    """interaction.py:557-577 with reduce_sum=True.  E [B, F, D] -> [B, P]; pair order i<j, i outer."""
    F = E.shape[1]
    row, col = zip(*itertools.combinations(range(F), 2)) if F > 1 else ((), ())
    row, col = list(row), list(col)
    return (E[:, row, :] * E[:, col, :]).sum(axis=2), (row, col)


def inner_product_backward(E, pairs, gp):
    row, col = pairs
    gE = np.zeros_like(E)
    np.add.at(gE, (slice(None), row), gp[:, :, None] * E[:, col, :])
    np.add.at(gE, (slice(None), col), gp[:, :, None] * E[:, row, :])
    return gE


def crossnet_forward(x0, kernels, bias, param):
    """interaction.py:438-453.  kernels [L, W, 1|W], bias [L, W, 1]."""
    xs = [x0]
    xl = x0
    for i in range(kernels.shape[0]):
        if param == "vector":
            s = xl @ kernels[i]                                   # [B, 1]
            xl = x0 * s + bias[i][:, 0] + xl
        else:
            u = xl @ kernels[i].T + bias[i][:, 0]                 # (W x_l + b)
            xl = x0 * u + xl
        xs.append(xl)
    return xl, xs


def crossnet_backward(g, xs, kernels, bias, param):
    x0 = xs[0]
    L = kernels.shape[0]
    gk, gb = np.zeros_like(kernels), np.zeros_like(bias)
    gx0 = np.zeros_like(x0)
    for i in reversed(range(L)):
        xl = xs[i]
        if param == "vector":
            s = xl @ kernels[i]
            c = (x0 * g).sum(axis=1, keepdims=True)              # [B, 1]
            gx0 += g * s
            gk[i] = (xl * c).sum(axis=0)[:, None]
            gb[i] = g.sum(axis=0)[:, None]
            g = g + c * kernels[i][:, 0]
        else:
            u = xl @ kernels[i].T + bias[i][:, 0]
            gu = g * x0
            gx0 += g * u
            gk[i] = gu.T @ xl
            gb[i] = gu.sum(axis=0)[:, None]
            g = g + gu @ kernels[i]
    return g + gx0, gk, gb


def cin_layer_forward(H, X0, W, b, relu=True):
    """One CIN layer (interaction.py:216-229): Z = einsum('bhd,bmd->bhmd', H, X0) flattened over (h, m), the 1x1 Conv1d
    W [O, h m] (+ bias), the activation.  H [B, h, D], X0 [B, m, D] -> (A [B, O, D], (Z, Y))."""
    B, _, D = X0.shape
    Z = (H[:, :, None, :] * X0[:, None, :, :]).reshape(B, H.shape[1] * X0.shape[1], D)
    Y = np.einsum("ok,bkd->bod", W, Z)
    if b is not None:
        Y = Y + b[None, :, None]
    return (np.maximum(Y, 0) if relu else Y), (Z, Y)


def cin_layer_backward(gA, H, X0, W, cache, relu=True):
    """Gradients of one CIN layer: (gH, gX0 through the products only, gW [O, h m], gb [O])."""
    Z, Y = cache
    B, F, D = X0.shape
    gY = gA * (Y > 0) if relu else gA
    gW = np.einsum("bod,bkd->ok", gY, Z)
    gb = gY.sum(axis=(0, 2))
    gZ = np.einsum("ok,bod->bkd", W, gY).reshape(B, H.shape[1], F, D)
    gH = (gZ * X0[:, None, :, :]).sum(axis=2)
    gX0 = (gZ * H[:, :, None, :]).sum(axis=1)
    return gH, gX0, gW, gb


def cin_forward(X0, P, prefix, layer_size, split_half, activation="relu"):
    """interaction.py:207-248.  X0 [B, F, D] -> [B, featuremap_num]."""
    hidden, finals, cache = [X0], [], []
    for i, size in enumerate(layer_size):
        H = hidden[-1]
        W, b = P[prefix + "conv1ds.%d.weight" % i][:, :, 0], P[prefix + "conv1ds.%d.bias" % i]
        A, (Z, Y) = cin_layer_forward(H, X0, W, b, activation == "relu")
        if split_half and i != len(layer_size) - 1:
            nxt, direct = A[:, :size // 2], A[:, size // 2:]
        elif split_half:
            nxt, direct = None, A
        else:
            nxt, direct = A, A
        finals.append(direct)
        hidden.append(nxt)
        cache.append((H, Z, Y))
    return np.concatenate(finals, axis=1).sum(axis=-1), cache


def cin_backward(gp, X0, cache, P, prefix, layer_size, split_half, grads, activation="relu"):
    B, F, D = X0.shape
    # split gp ([B, featuremap_num]) back to the per-layer direct parts
    widths = []
    for i, size in enumerate(layer_size):
        widths.append(size // 2 if (split_half and i != len(layer_size) - 1) else size)
    offs = np.cumsum([0] + widths)
    gX0 = np.zeros_like(X0)
    g_next = None
    for i in reversed(range(len(layer_size))):
        H, Z, Y = cache[i]
        gdir = np.repeat(gp[:, offs[i]:offs[i + 1], None], D, axis=2)
        if split_half and i != len(layer_size) - 1:
            gA = np.concatenate([g_next, gdir], axis=1)
        elif split_half:
            gA = gdir
        else:
            gA = gdir + (g_next if g_next is not None else 0)
        W = P[prefix + "conv1ds.%d.weight" % i][:, :, 0]
        gH, gX0_i, gW, gb = cin_layer_backward(gA, H, X0, W, (Z, Y), activation == "relu")
        grads[prefix + "conv1ds.%d.weight" % i] = gW[:, :, None]
        grads[prefix + "conv1ds.%d.bias" % i] = gb
        gX0 += gX0_i
        if i == 0:
            gX0 += gH
        else:
            g_next = gH
    return gX0


def senet_forward(E, W1, W2):
    """interaction.py:93-101."""
    Z = E.mean(axis=-1)
    A1 = np.maximum(Z @ W1.T, 0)
    A = np.maximum(A1 @ W2.T, 0)
    return E * A[:, :, None], (Z, A1, A)


def senet_backward(gV, E, cache, W1, W2):
    Z, A1, A = cache
    gE = gV * A[:, :, None]
    gA = (gV * E).sum(axis=-1) * (A > 0)
    gW2 = gA.T @ A1
    gA1 = (gA @ W2) * (A1 > 0)
    gW1 = gA1.T @ Z
    gZ = gA1 @ W1
    gE = gE + gZ[:, :, None] / E.shape[-1]
    return gE, gW1, gW2


def _bilinear_weights(P, prefix, btype, F):
    pairs = list(itertools.combinations(range(F), 2))
    if btype == "all":
        return pairs, [P[prefix + "bilinear.weight"]] * len(pairs), ["bilinear.weight"] * len(pairs)
    if btype == "each":
        return pairs, [P[prefix + "bilinear.%d.weight" % i] for i, _ in pairs], \
            ["bilinear.%d.weight" % i for i, _ in pairs]
    return pairs, [P[prefix + "bilinear.%d.weight" % k] for k in range(len(pairs))], \
        ["bilinear.%d.weight" % k for k in range(len(pairs))]


def bilinear_forward(V, P, prefix, btype):
    """interaction.py:140-156.  p_k = (v_i W_k^T) * v_j."""
    pairs, Ws, _ = _bilinear_weights(P, prefix, btype, V.shape[1])
    out = [(V[:, i, :] @ W.T) * V[:, j, :] for (i, j), W in zip(pairs, Ws)]
    return np.stack(out, axis=1) if out else np.zeros((V.shape[0], 0, V.shape[2]), V.dtype)


def bilinear_backward(gp, V, P, prefix, btype, grads):
    pairs, Ws, names = _bilinear_weights(P, prefix, btype, V.shape[1])
    gV = np.zeros_like(V)
    for k, ((i, j), W, name) in enumerate(zip(pairs, Ws, names)):
        t = V[:, i, :] @ W.T
        gV[:, j, :] += gp[:, k, :] * t
        gt = gp[:, k, :] * V[:, j, :]
        gV[:, i, :] += gt @ W
        key = prefix + name
        grads[key] = grads.get(key, 0) + gt.T @ V[:, i, :]
    return gV


def interacting_forward(E, P, prefix, head_num, use_res=True, scaling=False):
    """InteractingLayer (interaction.py:366-394): multi-head self-attention over the fields.  E [B, F, D]."""
    B, F, D = E.shape
    A = D // head_num
    q, k, v = E @ P[prefix + "W_Query"], E @ P[prefix + "W_key"], E @ P[prefix + "W_Value"]
    split = lambda t: t.reshape(B, F, head_num, A).transpose(0, 2, 1, 3)      # [B, H, F, A]  # noqa: E731
    qh, kh, vh = split(q), split(k), split(v)
    inner = qh @ kh.transpose(0, 1, 3, 2)
    if scaling:
        inner = inner / (A ** 0.5)
    ex = np.exp(inner - inner.max(axis=-1, keepdims=True))
    att = ex / ex.sum(axis=-1, keepdims=True)
    res = (att @ vh).transpose(0, 2, 1, 3).reshape(B, F, D)
    if use_res:
        res = res + E @ P[prefix + "W_Res"]
    return np.maximum(res, 0), (E, qh, kh, vh, att, res)


def interacting_backward(g, cache, P, prefix, head_num, grads, use_res=True, scaling=False):
    E, qh, kh, vh, att, pre = cache
    B, F, D = E.shape
    A = D // head_num
    gpre = g * (pre > 0)
    gE = np.zeros_like(E)
    if use_res:
        grads[prefix + "W_Res"] = np.einsum("bfd,bfe->de", E, gpre)
        gE += gpre @ P[prefix + "W_Res"].T
    go = gpre.reshape(B, F, head_num, A).transpose(0, 2, 1, 3)                  # [B, H, F, A]
    gatt = go @ vh.transpose(0, 1, 3, 2)
    gv = att.transpose(0, 1, 3, 2) @ go
    gs = att * (gatt - (att * gatt).sum(axis=-1, keepdims=True))
    if scaling:
        gs = gs / (A ** 0.5)
    gq, gk = gs @ kh, gs.transpose(0, 1, 3, 2) @ qh
    join = lambda t: t.transpose(0, 2, 1, 3).reshape(B, F, D)                   # noqa: E731
    for name, gt in (("W_Query", join(gq)), ("W_key", join(gk)), ("W_Value", join(gv))):
        grads[prefix + name] = np.einsum("bfd,bfe->de", E, gt)
        gE += gt @ P[prefix + name].T
    return gE


def crossnet_mix_forward(x0, P, prefix, n_experts):
    """CrossNetMix (interaction.py:499-534).  x0 [B, W]."""
    U, V, C, bias = P[prefix + "U_list"], P[prefix + "V_list"], P[prefix + "C_list"], P[prefix + "bias"]
    gw = np.concatenate([P[prefix + "gating.%d.weight" % e] for e in range(n_experts)], axis=0)   # [E, W]
    xl, caches = x0, []
    for i in range(U.shape[0]):
        s = xl @ gw.T
        ex = np.exp(s - s.max(axis=1, keepdims=True))
        score = ex / ex.sum(axis=1, keepdims=True)                              # [B, E]
        v1 = np.tanh(np.einsum("bw,ewr->ber", xl, V[i]))
        v2 = np.tanh(np.einsum("ers,bes->ber", C[i], v1))
        uv = np.einsum("ewr,ber->bew", U[i], v2)
        dot = x0[:, None, :] * (uv + bias[i][:, 0])
        caches.append((xl, score, v1, v2, uv, dot))
        xl = np.einsum("bew,be->bw", dot, score) + xl
    return xl, (x0, gw, caches)


def crossnet_mix_backward(g, cache, P, prefix, n_experts, grads):
    x0, gw, caches = cache
    U, V, C, bias = P[prefix + "U_list"], P[prefix + "V_list"], P[prefix + "C_list"], P[prefix + "bias"]
    gU, gV, gC, gb = np.zeros_like(U), np.zeros_like(V), np.zeros_like(C), np.zeros_like(bias)
    ggw = np.zeros_like(gw)
    gx0 = np.zeros_like(x0)
    gxl = g
    for i in reversed(range(U.shape[0])):
        xl, score, v1, v2, uv, dot = caches[i]
        gdot = gxl[:, None, :] * score[:, :, None]                              # [B, E, W]
        gscore = np.einsum("bew,bw->be", dot, gxl)
        gs = score * (gscore - (score * gscore).sum(axis=1, keepdims=True))
        ggw += gs.T @ xl
        gprev = gxl + gs @ gw
        gx0 += (gdot * (uv + bias[i][:, 0])).sum(axis=1)
        guv = gdot * x0[:, None, :]
        gb[i][:, 0] = guv.sum(axis=(0, 1))
        gU[i] = np.einsum("bew,ber->ewr", guv, v2)
        gv2 = np.einsum("ewr,bew->ber", U[i], guv) * (1 - v2 * v2)
        gC[i] = np.einsum("ber,bes->ers", gv2, v1)
        gv1 = np.einsum("ers,ber->bes", C[i], gv2) * (1 - v1 * v1)
        gV[i] = np.einsum("bw,ber->ewr", xl, gv1)
        gprev = gprev + np.einsum("ewr,ber->bw", V[i], gv1)
        gxl = gprev
    grads[prefix + "U_list"], grads[prefix + "V_list"], grads[prefix + "C_list"], grads[prefix + "bias"] = gU, gV, gC, gb
    for e in range(n_experts):
        grads[prefix + "gating.%d.weight" % e] = ggw[e:e + 1]
    return gxl + gx0


def linear_forward(X, columns, fi, P):
    """Linear.forward, basemodel.py:63-92."""
    embs, cache = lookup_forward(X, columns, fi, P, "linear_model.embedding_dict.")
    logit = np.zeros((X.shape[0], 1), X.dtype)
    if embs:
        logit = logit + np.concatenate(embs, axis=1).sum(axis=1, keepdims=True)
    _, _, dense = _split(columns)
    dense_x = None
    if dense:
        dense_x = np.concatenate([X[:, fi[c["name"]][0]:fi[c["name"]][1]] for c in dense], axis=1)
        logit = logit + dense_x @ P["linear_model.weight"]
    return logit, (cache, dense_x)


def linear_backward(g, columns, cache, P, grads):
    emb_cache, dense_x = cache
    lookup_backward([g] * len(emb_cache), emb_cache, P, "linear_model.embedding_dict.", grads)
    if dense_x is not None:
        grads["linear_model.weight"] = dense_x.T @ g


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bce_sum(y_pred, y_true):
    """F.binary_cross_entropy(reduction='sum') with its log clamp at -100.  basemodel.py:254."""
    lp = np.maximum(np.log(y_pred), -100.0)
    l1p = np.maximum(np.log1p(-y_pred), -100.0)
    return float(-(y_true * lp + (1 - y_true) * l1p).sum())


# --------------------------------------------------------------------------------------------------
# whole models: forward to the pre-sigmoid logit, backward from d loss / d logit
# --------------------------------------------------------------------------------------------------
class Oracle(object):
    def __init__(self, spec, params, dtype=np.float32):
        self.spec = spec
        self.kw = spec.get("kwargs", {})
        self.dt = np.dtype(dtype)
        self.P = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
        self.lin_cols, self.dnn_cols = spec["linear_columns"], spec["dnn_columns"]
        self.fi = build_input_features(self.lin_cols + self.dnn_cols)
        self.model = spec["model"]
        hidden = self.kw.get("dnn_hidden_units", None)
        self.n_dnn = len([k for k in self.P if k.startswith("dnn.linears.") and k.endswith(".weight")]) \
            if hidden is None else len(hidden)

    # -- forward -------------------------------------------------------------------------------------
    def forward(self, X):
        X = np.asarray(X, dtype=self.dt)
        P, fi, kw = self.P, self.fi, self.kw
        c = {"X": X}
        embs, c["emb_cache"] = lookup_forward(X, self.dnn_cols, fi, P, "embedding_dict.")
        _, _, dense = _split(self.dnn_cols)
        dense_x = np.concatenate([X[:, fi[d["name"]][0]:fi[d["name"]][1]] for d in dense], axis=1) \
            if dense else np.zeros((X.shape[0], 0), self.dt)
        E = np.stack(embs, axis=1) if embs and len(set(e.shape[1] for e in embs)) == 1 else None
        c["E"], c["embs"], c["n_dense"] = E, embs, dense_x.shape[1]
        flat = np.concatenate(embs, axis=1) if embs else np.zeros((X.shape[0], 0), self.dt)
        c["emb_width"] = flat.shape[1]
        lin, c["lin_cache"] = linear_forward(X, self.lin_cols, fi, P)
        logit = np.zeros((X.shape[0], 1), self.dt)
        m = self.model
        if m in ("DeepFM", "WDL"):       # WDL (wdl.py:57-75) = DeepFM without the FM term
            logit = logit + lin
            c["use_fm"] = m == "DeepFM" and kw.get("use_fm", True) and len(embs) > 0
            if c["use_fm"]:
                logit = logit + fm_forward(E)
            c["use_dnn"] = self.n_dnn > 0 and len(self.dnn_cols) > 0
            if c["use_dnn"]:
                h, c["acts"] = dnn_forward(np.concatenate([flat, dense_x], axis=1), P, "dnn.", self.n_dnn)
                logit = logit + h @ P["dnn_linear.weight"].T
        elif m == "xDeepFM":
            logit = logit + lin
            ls = tuple(kw.get("cin_layer_size", (256, 128)))
            c["use_cin"] = len(ls) > 0 and len(self.dnn_cols) > 0
            if c["use_cin"]:
                co, c["cin_cache"] = cin_forward(E, P, "cin.", ls, kw.get("cin_split_half", True),
                                                 kw.get("cin_activation", "relu"))
                c["cin_out"] = co
                logit = logit + co @ P["cin_linear.weight"].T
            c["use_dnn"] = self.n_dnn > 0 and len(self.dnn_cols) > 0
            if c["use_dnn"]:
                h, c["acts"] = dnn_forward(np.concatenate([flat, dense_x], axis=1), P, "dnn.", self.n_dnn)
                logit = logit + h @ P["dnn_linear.weight"].T
        elif m == "FiBiNET":
            V, c["se_cache"] = senet_forward(E, P["SE.excitation.0.weight"], P["SE.excitation.2.weight"])
            bt = kw.get("bilinear_type", "interaction")
            pv, pe = bilinear_forward(V, P, "Bilinear.", bt), bilinear_forward(E, P, "Bilinear.", bt)
            c["V"] = V
            both = np.concatenate([pv, pe], axis=1)
            c["n_pairs"] = pv.shape[1]
            h, c["acts"] = dnn_forward(np.concatenate([both.reshape(X.shape[0], -1), dense_x], axis=1), P, "dnn.",
                                       self.n_dnn)
            dnn_logit = h @ P["dnn_linear.weight"].T
            if len(self.lin_cols) > 0 and len(self.dnn_cols) > 0:
                logit = lin + dnn_logit
            elif len(self.lin_cols) == 0:
                logit = dnn_logit
            else:
                logit = lin
        elif m == "DCN":
            logit = logit + lin
            x0 = np.concatenate([flat, dense_x], axis=1)
            cross_num = kw.get("cross_num", 2)
            parts = []
            if cross_num > 0:
                co, c["cross_xs"] = crossnet_forward(x0, P["crossnet.kernels"], P["crossnet.bias"],
                                                     kw.get("cross_parameterization", "vector"))
                parts.append(co)
            if self.n_dnn > 0:
                h, c["acts"] = dnn_forward(x0, P, "dnn.", self.n_dnn)
                parts.append(h)
            c["stack_split"] = parts[0].shape[1] if len(parts) == 2 else None
            if parts:
                c["stack"] = np.concatenate(parts, axis=1)
                logit = logit + c["stack"] @ P["dnn_linear.weight"].T
        elif m == "AutoInt":   # autoint.py:80-112
            logit = logit + lin
            n_att = len([k for k in P if k.startswith("int_layers.") and k.endswith("W_Query")])
            heads = kw.get("att_head_num", 2)
            att, c["att_caches"] = E, []
            for l in range(n_att):
                att, cc = interacting_forward(att, P, "int_layers.%d." % l, heads, kw.get("att_res", True))
                c["att_caches"].append(cc)
            parts = []
            if n_att > 0:
                parts.append(att.reshape(X.shape[0], -1))
            if self.n_dnn > 0:
                h, c["acts"] = dnn_forward(np.concatenate([flat, dense_x], axis=1), P, "dnn.", self.n_dnn)
                parts.append(h)
            c["stack_split"] = parts[0].shape[1] if len(parts) == 2 else None
            c["n_att"] = n_att
            c["stack"] = np.concatenate(parts, axis=1)
            logit = logit + c["stack"] @ P["dnn_linear.weight"].T
        elif m == "DCNMix":    # dcnmix.py:80-100
            logit = logit + lin
            x0 = np.concatenate([flat, dense_x], axis=1)
            cross_num = kw.get("cross_num", 2)
            parts = []
            if cross_num > 0:
                co, c["mix_cache"] = crossnet_mix_forward(x0, P, "crossnet.", kw.get("num_experts", 4))
                parts.append(co)
            if self.n_dnn > 0:
                h, c["acts"] = dnn_forward(x0, P, "dnn.", self.n_dnn)
                parts.append(h)
            c["stack_split"] = parts[0].shape[1] if len(parts) == 2 else None
            if parts:
                c["stack"] = np.concatenate(parts, axis=1)
                logit = logit + c["stack"] @ P["dnn_linear.weight"].T
        elif m == "AFM":     # afm.py:60-75: linear + AFMLayer (interaction.py:299-325), or FM without attention
            logit = logit + lin
            c["use_att"] = kw.get("use_attention", True)
            if E is not None and len(embs) > 0:
                if c["use_att"]:
                    F_ = E.shape[1]
                    ii, jj = np.triu_indices(F_, 1)
                    bi = E[:, ii] * E[:, jj]                                         # [B, P, D]
                    pre = bi @ P["fm.attention_W"] + P["fm.attention_b"]
                    t = np.maximum(pre, 0)
                    s = t @ P["fm.projection_h"]                                    # [B, P, 1]
                    ex = np.exp(s - s.max(axis=1, keepdims=True))
                    a = ex / ex.sum(axis=1, keepdims=True)
                    o = (a * bi).sum(axis=1)                                         # [B, D]
                    c["afm"] = (ii, jj, bi, pre, t, a, o)
                    logit = logit + o @ P["fm.projection_p"]
                else:
                    logit = logit + fm_forward(E)
        elif m == "NFM":     # nfm.py:60-80: linear + DNN([BiInteractionPooling(E) | dense])
            s = E.sum(axis=1)
            bi = 0.5 * (s * s - (E * E).sum(axis=1))                       # interaction.py:54-61
            h, c["acts"] = dnn_forward(np.concatenate([bi, dense_x], axis=1), P, "dnn.", self.n_dnn)
            logit = lin + h @ P["dnn_linear.weight"].T
        elif m == "PNN":
            prods = []
            c["n_ip"] = 0
            if kw.get("use_inner", True):
                ip, c["pairs"] = inner_product_forward(E)
                c["n_ip"] = ip.shape[1]
                prods.append(ip)
            c["n_op"] = 0
            if kw.get("use_outter", False):          # OutterProductLayer, interaction.py:616-670
                ii, jj = np.triu_indices(E.shape[1], 1)
                p_, q_ = E[:, ii], E[:, jj]
                K = P["outterproduct.kernel"]
                kt = kw.get("kernel_type", "mat")
                op = np.einsum("bke,fke,bkf->bk", p_, K, q_) if kt == "mat" else (p_ * q_ * K[None]).sum(axis=-1)
                c["op"] = (ii, jj, p_, q_, kt)
                c["n_op"] = op.shape[1]
                prods.append(op)
            h, c["acts"] = dnn_forward(np.concatenate([flat] + prods + [dense_x], axis=1), P, "dnn.", self.n_dnn)
            logit = h @ P["dnn_linear.weight"].T
        else:
            raise ValueError(m)
        c["logit"] = logit
        self.cache = c
        out = logit + P["out.bias"]
        return logit, (sigmoid(out) if kw.get("task", "binary") == "binary" else out)

    # -- backward ------------------------------------------------------------------------------------
    def backward(self, g_logit):
        """g_logit = d loss / d (pre-bias logit), [B, 1].  Returns {state_dict key: gradient}."""
        P, c, kw, m = self.P, self.cache, self.kw, self.model
        g = np.asarray(g_logit, dtype=self.dt).reshape(-1, 1)
        grads = {"out.bias": g.sum(axis=0)}
        B = g.shape[0]
        W_emb = c["emb_width"]
        g_flat = np.zeros((B, W_emb), self.dt)
        E = c["E"]
        g_lin = None

        def dnn_head(g_logit_part, name="dnn."):
            acts = c["acts"]
            grads["dnn_linear.weight"] = g_logit_part.T @ acts[-1]
            return dnn_backward(g_logit_part @ P["dnn_linear.weight"], acts, P, name, self.n_dnn, grads)

        if m in ("DeepFM", "WDL"):
            g_lin = g
            if c["use_fm"]:
                g_flat += fm_backward(E, g).reshape(B, -1)
            if c["use_dnn"]:
                g_flat += dnn_head(g)[:, :W_emb]
        elif m == "xDeepFM":
            g_lin = g
            if c["use_cin"]:
                grads["cin_linear.weight"] = g.T @ c["cin_out"]
                gp = g @ P["cin_linear.weight"]
                g_flat += cin_backward(gp, E, c["cin_cache"], P, "cin.", tuple(kw.get("cin_layer_size", (256, 128))),
                                       kw.get("cin_split_half", True), grads,
                                       kw.get("cin_activation", "relu")).reshape(B, -1)
            if c["use_dnn"]:
                g_flat += dnn_head(g)[:, :W_emb]
        elif m == "FiBiNET":
            has_lin, has_dnn = len(self.lin_cols) > 0, len(self.dnn_cols) > 0
            g_lin = g if has_lin else None
            if has_dnn or not has_lin:
                gin = dnn_head(g)
                npairs, D = c["n_pairs"], E.shape[2]
                gboth = gin[:, :2 * npairs * D].reshape(B, 2 * npairs, D)
                bt = kw.get("bilinear_type", "interaction")
                gV = bilinear_backward(gboth[:, :npairs], c["V"], P, "Bilinear.", bt, grads)
                gE = bilinear_backward(gboth[:, npairs:], E, P, "Bilinear.", bt, grads)
                gE2, gW1, gW2 = senet_backward(gV, E, c["se_cache"], P["SE.excitation.0.weight"],
                                               P["SE.excitation.2.weight"])
                grads["SE.excitation.0.weight"], grads["SE.excitation.2.weight"] = gW1, gW2
                g_flat += (gE + gE2).reshape(B, -1)
        elif m == "DCN":
            g_lin = g
            if "stack" in c:
                grads["dnn_linear.weight"] = g.T @ c["stack"]
                gs = g @ P["dnn_linear.weight"]
                sp = c["stack_split"]
                cross_num = kw.get("cross_num", 2)
                g_cross = gs[:, :sp] if sp is not None else (gs if cross_num > 0 else None)
                g_deep = gs[:, sp:] if sp is not None else (gs if cross_num == 0 else None)
                gx0 = np.zeros((B, W_emb + c["n_dense"]), self.dt)
                if g_cross is not None:
                    gx, gk, gb = crossnet_backward(g_cross, c["cross_xs"], P["crossnet.kernels"], P["crossnet.bias"],
                                                   kw.get("cross_parameterization", "vector"))
                    grads["crossnet.kernels"], grads["crossnet.bias"] = gk, gb
                    gx0 += gx
                if g_deep is not None:
                    gx0 += dnn_backward(g_deep, c["acts"], P, "dnn.", self.n_dnn, grads)
                g_flat += gx0[:, :W_emb]
        elif m == "AutoInt":
            g_lin = g
            grads["dnn_linear.weight"] = g.T @ c["stack"]
            gs = g @ P["dnn_linear.weight"]
            sp, n_att = c["stack_split"], c["n_att"]
            g_att = gs[:, :sp] if sp is not None else (gs if n_att > 0 else None)
            g_deep = gs[:, sp:] if sp is not None else (gs if n_att == 0 else None)
            if g_att is not None:
                ga = g_att.reshape(E.shape)
                for l in reversed(range(n_att)):
                    ga = interacting_backward(ga, c["att_caches"][l], P, "int_layers.%d." % l, kw.get("att_head_num", 2),
                                              grads, kw.get("att_res", True))
                g_flat += ga.reshape(B, -1)
            if g_deep is not None:
                g_flat += dnn_backward(g_deep, c["acts"], P, "dnn.", self.n_dnn, grads)[:, :W_emb]
        elif m == "DCNMix":
            g_lin = g
            if "stack" in c:
                grads["dnn_linear.weight"] = g.T @ c["stack"]
                gs = g @ P["dnn_linear.weight"]
                sp = c["stack_split"]
                cross_num = kw.get("cross_num", 2)
                g_cross = gs[:, :sp] if sp is not None else (gs if cross_num > 0 else None)
                g_deep = gs[:, sp:] if sp is not None else (gs if cross_num == 0 else None)
                gx0 = np.zeros((B, W_emb + c["n_dense"]), self.dt)
                if g_cross is not None:
                    gx0 += crossnet_mix_backward(g_cross, c["mix_cache"], P, "crossnet.", kw.get("num_experts", 4), grads)
                if g_deep is not None:
                    gx0 += dnn_backward(g_deep, c["acts"], P, "dnn.", self.n_dnn, grads)
                g_flat += gx0[:, :W_emb]
        elif m == "AFM":
            g_lin = g
            if E is not None and len(c["embs"]) > 0:
                if c["use_att"]:
                    ii, jj, bi, pre, t, a, o = c["afm"]
                    grads["fm.projection_p"] = o.T @ g
                    go = g @ P["fm.projection_p"].T                                  # [B, D]
                    ga = (bi * go[:, None, :]).sum(axis=2, keepdims=True)            # [B, P, 1]
                    gbi = a * go[:, None, :]
                    gs = a * (ga - (a * ga).sum(axis=1, keepdims=True))
                    grads["fm.projection_h"] = np.einsum("bpa,bpo->ao", t, gs)
                    gpre = (gs @ P["fm.projection_h"].T) * (pre > 0)
                    grads["fm.attention_b"] = gpre.sum(axis=(0, 1))
                    grads["fm.attention_W"] = np.einsum("bpd,bpa->da", bi, gpre)
                    gbi = gbi + gpre @ P["fm.attention_W"].T
                    gE = np.zeros_like(E)
                    np.add.at(gE, (slice(None), ii), gbi * E[:, jj])
                    np.add.at(gE, (slice(None), jj), gbi * E[:, ii])
                    g_flat += gE.reshape(B, -1)
                else:
                    g_flat += fm_backward(E, g).reshape(B, -1)
        elif m == "NFM":
            g_lin = g
            gin = dnn_head(g)
            D = E.shape[2]
            g_flat += (gin[:, None, :D] * (E.sum(axis=1, keepdims=True) - E)).reshape(B, -1)
        elif m == "PNN":
            gin = dnn_head(g)
            g_flat += gin[:, :W_emb]
            if c["n_ip"]:
                gip = gin[:, W_emb:W_emb + c["n_ip"]]
                g_flat += inner_product_backward(E, c["pairs"], gip).reshape(B, -1)
            if c["n_op"]:
                ii, jj, p_, q_, kt = c["op"]
                gop = gin[:, W_emb + c["n_ip"]:W_emb + c["n_ip"] + c["n_op"]]
                K = P["outterproduct.kernel"]
                if kt == "mat":
                    grads["outterproduct.kernel"] = np.einsum("bk,bke,bkf->fke", gop, p_, q_)
                    gp = np.einsum("bk,fke,bkf->bke", gop, K, q_)
                    gq = np.einsum("bk,bke,fke->bkf", gop, p_, K)
                else:
                    gk = (gop[:, :, None] * p_ * q_).sum(axis=0)
                    grads["outterproduct.kernel"] = gk if kt == "vec" else gk.sum(axis=1, keepdims=True)
                    gp, gq = gop[:, :, None] * q_ * K[None], gop[:, :, None] * p_ * K[None]
                gE = np.zeros_like(E)
                np.add.at(gE, (slice(None), ii), gp)
                np.add.at(gE, (slice(None), jj), gq)
                g_flat += gE.reshape(B, -1)

        # embedding tables
        g_embs, off = [], 0
        for e in c["embs"]:
            g_embs.append(g_flat[:, off:off + e.shape[1]])
            off += e.shape[1]
        lookup_backward(g_embs, c["emb_cache"], P, "embedding_dict.", grads)
        if g_lin is not None:
            linear_backward(g_lin, self.lin_cols, c["lin_cache"], P, grads)
        for k in P:  # parameters the batch never touched still have a (zero) dense gradient
            if k.endswith(".weight") and ("embedding_dict." in k) and k not in grads:
                grads[k] = np.zeros_like(P[k])
        return grads

    # -- one reference training step (basemodel.py:242-262), l2 = 0 ------------------------------------
    def train_step(self, X, y, optimizer="sgd", lr=0.01, eps=1e-10, state=None):
        """loss = BCE(sum); dense SGD / Adagrad update of EVERY parameter, as torch.optim does."""
        logit, y_pred = self.forward(X)
        y = np.asarray(y, dtype=self.dt).reshape(-1, 1)
        loss = bce_sum(y_pred.astype(np.float64), y.astype(np.float64))
        grads = self.backward(y_pred - y)
        state = {} if state is None else state
        for k, g in grads.items():
            g = np.asarray(g, dtype=self.dt).reshape(self.P[k].shape)
            self.P[k], s = optimizer_step(optimizer, self.P[k], g, state.get(k), self.dt.type(lr), self.dt.type(eps))
            if s is not None:
                state[k] = s
        return loss, state
