"""GPU parity tests of the fused gather / scatter path (csrc/embed.hip, fm.hip) through the drop-in models.

Checker = committed golden vectors from the real reference + the numpy oracle.  Tolerances:
  logits / predictions  1e-5 absolute (north_star)
  gradients             2e-5 x max|reference gradient| (fp32 re-association over the batch)
  3-step trajectories   2e-5 absolute on every parameter
"""
import numpy as np
import pytest
import torch

from helpers import build_model, golden_names, load_golden, max_abs
from np_oracle import Oracle, fm_backward, fm_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL, GRAD_TOL, TRAJ_TOL = 1e-5, 2e-5, 2e-5


def _loaded(name, l2=0.0):
    g = load_golden(name)
    m = build_model(g["spec"], DEV, l2=l2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    return g, m


def _report(tag, got, ref):
    got, ref = np.asarray(got, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    err = np.abs(got - ref)
    i = int(err.argmax()) if err.size else 0
    return "%s: max|d|=%.3e at %d (got %.7g ref %.7g) mean|d|=%.3e" % (
        tag, err.max() if err.size else 0, i, got[i] if err.size else 0, ref[i] if err.size else 0,
        err.mean() if err.size else 0)


DEEPFM = golden_names("deepfm")


@pytest.mark.parametrize("name", DEEPFM)
def test_forward_logits_match_reference(name):
    g, m = _loaded(name)
    m.eval()
    cap = {}
    h = m.out.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("logit", inp[0].detach()))
    with torch.no_grad():
        y = m(torch.from_numpy(g["X"]).to(DEV))
    h.remove()
    torch.cuda.synchronize()
    m.model_plan().check_ids()
    assert max_abs(cap["logit"].cpu().numpy(), g["logit"]) <= LOGIT_TOL, _report("logit", cap["logit"].cpu().numpy(), g["logit"])
    assert max_abs(y.cpu().numpy(), g["y_pred"]) <= LOGIT_TOL
    # and against the oracle evaluated in fp64
    l64, _ = Oracle(g["spec"], g["params"], dtype=np.float64).forward(g["X"])
    assert max_abs(cap["logit"].cpu().numpy(), l64) <= LOGIT_TOL


@pytest.mark.parametrize("name", DEEPFM)
def test_fused_inputs_match_oracle_piecewise(name):
    """dnn_input layout, linear logit and FM term individually (not just their sum)."""
    g, m = _loaded(name)
    o = Oracle(g["spec"], g["params"], dtype=np.float64)
    o.forward(g["X"])
    c = o.cache
    plan = m.model_plan()
    want_fm = c.get("use_fm", False)
    with torch.no_grad():
        out, wide, fm = m.fused_inputs(torch.from_numpy(g["X"]).to(DEV), want_fm=want_fm)
    flat = np.concatenate(c["embs"], axis=1) if c["embs"] else np.zeros((g["X"].shape[0], 0))
    assert out.shape[1] == plan.width
    assert max_abs(out[:, :flat.shape[1]].cpu().numpy(), flat) <= 1e-6, _report("emb", out[:, :flat.shape[1]].cpu().numpy(), flat)
    if plan.dense_cols:
        dense = g["X"][:, plan.dense_cols]
        assert max_abs(out[:, flat.shape[1]:].cpu().numpy(), dense) == 0.0
    from np_oracle import linear_forward
    lin, _ = linear_forward(np.asarray(g["X"], np.float64), o.lin_cols, o.fi, o.P)
    assert max_abs(wide.cpu().numpy(), lin) <= 2e-6, _report("wide", wide.cpu().numpy(), lin)
    if want_fm:
        assert max_abs(fm.cpu().numpy(), fm_forward(c["E"])) <= 5e-6, _report("fm", fm.cpu().numpy(), fm_forward(c["E"]))


@pytest.mark.parametrize("name", DEEPFM)
def test_dense_gradients_match_reference(name):
    """Default update mode: param.grad of every parameter (tables included) equals the reference's."""
    g, m = _loaded(name)
    m.train()
    assert m.model_plan().update == ("dense",)
    X, y = torch.from_numpy(g["X"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    y_pred = m(X).squeeze()
    loss = torch.nn.functional.binary_cross_entropy(y_pred, y, reduction="sum")
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - g["loss"]) <= 1e-4 * max(1.0, abs(g["loss"]))
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        scale = max(1.0, float(np.max(np.abs(ref))))
        assert max_abs(got, ref) <= GRAD_TOL * scale, _report(k, got, ref)
    # a second backward accumulates (autograd semantics), zero_grad + backward starts from zero again
    y_pred = m(X).squeeze()
    torch.nn.functional.binary_cross_entropy(y_pred, y, reduction="sum").backward()
    tables = [(k, p) for k, p in m.named_parameters() if "embedding_dict" in k]
    if not tables:      # dense-only model: nothing to scatter
        return
    k0, p0 = tables[0]
    assert max_abs(p0.grad.cpu().numpy(), 2 * g["grads"][k0]) <= 2 * GRAD_TOL * max(1.0, np.abs(g["grads"][k0]).max())
    m.zero_grad()
    y_pred = m(X).squeeze()
    torch.nn.functional.binary_cross_entropy(y_pred, y, reduction="sum").backward()
    assert max_abs(p0.grad.cpu().numpy(), g["grads"][k0]) <= GRAD_TOL * max(1.0, np.abs(g["grads"][k0]).max())


@pytest.mark.parametrize("name", [n for n in DEEPFM if "X_steps" in load_golden(n)["extra"]])
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_fused_sparse_training_matches_reference_trajectory(name, opt):
    """compile('sgd'|'adagrad') with l2=0: the O(batch) fused update reproduces 3 reference steps."""
    g, m = _loaded(name)
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    plan = m.model_plan()
    assert plan.update[0] == opt
    losses = []
    for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
        loss, _, _ = m._train_step(torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV))
        losses.append(loss.item())
    plan.check_ids()
    np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=2e-5)
    sd = m.state_dict()
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            key = k[len(opt) + 2:]
            assert max_abs(sd[key].cpu().numpy(), v) <= TRAJ_TOL, _report(key, sd[key].cpu().numpy(), v)
    # zero-at-rest invariant of the gradient slabs
    for p in plan.table_params:
        slab = plan.gacc_of(p)
        if slab is not None:
            assert float(slab.abs().max().item()) == 0.0


@pytest.mark.parametrize("opt", ["adam", "rmsprop"])
def test_dense_mode_with_any_torch_optimizer(opt):
    """Optimizers whose untouched rows move (momentum) take the exact dense-gradient path; with the
    reference's default L2 on every row too.  Compared with the oracle's dense gradient + torch's update."""
    g, m = _loaded("deepfm_mixed", l2=1e-5)
    m.compile(opt, "binary_crossentropy", metrics=[])
    assert m.model_plan().update == ("dense",)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    X, y = torch.from_numpy(g["X"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    m._train_step(X, y)
    # replay on CPU with torch: same optimizer, gradient = oracle dense grad + 2*l2*p for the regularised groups
    o = Oracle(g["spec"], {k: v.cpu().numpy() for k, v in before.items()}, dtype=np.float64)
    _, yp = o.forward(g["X"])
    grads = o.backward(yp - g["y"].reshape(-1, 1))
    cpu_params = {k: torch.nn.Parameter(v.cpu().clone()) for k, v in before.items()}
    optim = {"adam": torch.optim.Adam, "rmsprop": torch.optim.RMSprop}[opt](list(cpu_params.values()))
    for k, p in cpu_params.items():
        gk = torch.from_numpy(np.asarray(grads[k], np.float32).reshape(tuple(p.shape)))
        if "embedding_dict" in k or k.startswith("linear_model."):
            gk = gk + 2e-5 * p.detach()
        p.grad = gk
    optim.step()
    after = m.state_dict()
    for k, p in cpu_params.items():
        assert max_abs(after[k].cpu().numpy(), p.detach().numpy()) <= 5e-6, k


def test_fm_layer_matches_oracle():
    from deepctr_torch.layers import FM
    rng = np.random.default_rng(3)
    for (B, F, D) in [(37, 26, 16), (5, 3, 4), (64, 7, 5), (9, 2, 70)]:
        E = rng.normal(0, 0.5, (B, F, D)).astype(np.float32)
        t = torch.from_numpy(E).to(DEV).requires_grad_(True)
        y = FM()(t)
        assert y.shape == (B, 1)
        ref = fm_forward(E.astype(np.float64))
        assert max_abs(y.detach().cpu().numpy(), ref) <= 2e-5 * max(1, np.abs(ref).max())
        gy = rng.normal(0, 1, (B, 1)).astype(np.float32)
        y.backward(torch.from_numpy(gy).to(DEV))
        gref = fm_backward(E.astype(np.float64), gy.astype(np.float64))
        assert max_abs(t.grad.cpu().numpy(), gref) <= 2e-5 * max(1, np.abs(gref).max())
    with pytest.raises(ValueError):
        FM()(torch.zeros(3, 4, device=DEV))


def test_reference_shaped_accessors():
    """input_from_feature_columns / linear_model(X) / embedding_lookup return what the reference's do."""
    g, m = _loaded("deepfm_mixed")
    o = Oracle(g["spec"], g["params"], dtype=np.float64)
    o.forward(g["X"])
    X = torch.from_numpy(g["X"]).to(DEV)
    with torch.no_grad():
        embs, dense = m.input_from_feature_columns(X, m.dnn_feature_columns, m.embedding_dict)
        lin = m.linear_model(X)
    assert len(embs) == len(o.cache["embs"]) and all(e.shape[1] == 1 for e in embs)
    for e, ref in zip(embs, o.cache["embs"]):
        assert max_abs(e[:, 0].cpu().numpy(), ref) <= 1e-6
    assert [tuple(d.shape) for d in dense] == [(g["X"].shape[0], 1), (g["X"].shape[0], 3)]
    from np_oracle import linear_forward
    ref_lin, _ = linear_forward(np.asarray(g["X"], np.float64), o.lin_cols, o.fi, o.P)
    assert max_abs(lin.cpu().numpy(), ref_lin) <= 2e-6
    from deepctr_torch.inputs import varlen_embedding_lookup
    vcols = [c for c in m.dnn_feature_columns if hasattr(c, "maxlen")]
    seqs = varlen_embedding_lookup(X, m.embedding_dict, m.feature_index, vcols)
    for c in vcols:
        W = g["params"]["embedding_dict.%s.weight" % c.embedding_name]
        lo, hi = m.feature_index[c.name]
        ref = W[g["X"][:, lo:hi].astype(np.int64)]
        assert max_abs(seqs[c.name].detach().cpu().numpy(), ref) == 0.0


def test_linear_with_sparse_feat_refine_weight():
    """Linear.forward(X, sparse_feat_refine_weight=m_x) (IFM / DIFM; reference basemodel.py:63-92) against a plain torch
    restatement of the reference's lines: value, d / d m_x, and the table gradients (SGD step read back)."""
    g, m = _loaded("deepfm_mixed")
    lm = m.linear_model
    X = torch.from_numpy(g["X"]).to(DEV)
    B = X.shape[0]
    cols = list(lm.sparse_feature_columns) + list(lm.varlen_sparse_feature_columns)
    gen = torch.Generator().manual_seed(5)
    mx = (torch.rand(B, len(cols), generator=gen) + 0.5).to(DEV).requires_grad_(True)
    gy = torch.randn(B, 1, generator=gen).to(DEV)

    # the reference's operations on plain tensors (fp32, same order)
    tabs = {k: v.weight.detach().clone().requires_grad_(True) for k, v in lm.embedding_dict.items()}
    mx_ref = mx.detach().clone().requires_grad_(True)
    parts = []
    for fc in lm.sparse_feature_columns:
        lo, hi = lm.feature_index[fc.name]
        parts.append(tabs[fc.embedding_name][X[:, lo:hi].long()])                     # [B, 1, 1]
    for fc in lm.varlen_sparse_feature_columns:
        lo, hi = lm.feature_index[fc.name]
        ids = X[:, lo:hi].long()
        e = tabs[fc.embedding_name][ids]                                               # [B, T, 1]
        if fc.length_name is None:
            mask = (ids != 0).float().unsqueeze(-1)
        else:
            ln = X[:, lm.feature_index[fc.length_name][0]].long()
            mask = (torch.arange(ids.shape[1], device=DEV)[None, :] < ln[:, None]).float().unsqueeze(-1)
        if fc.combiner == "max":
            parts.append(torch.max(e - (1 - mask) * 1e9, dim=1, keepdim=True)[0])
        else:
            sm = torch.sum(e * mask, dim=1, keepdim=False)
            if fc.combiner == "mean":
                sm = sm / (mask.sum(dim=1) + 1e-8)
            parts.append(sm.unsqueeze(1))
    cat = torch.cat(parts, dim=-1) * mx_ref.unsqueeze(1)
    ref = torch.sum(cat, dim=-1, keepdim=False)
    dense = [X[:, lm.feature_index[fc.name][0]:lm.feature_index[fc.name][1]] for fc in lm.dense_feature_columns]
    w_ref = lm.weight.detach().clone().requires_grad_(True) if dense else None
    if dense:
        ref = ref + torch.cat(dense, dim=-1).matmul(w_ref)
    ref.backward(gy)

    m.compile("sgd", "binary_crossentropy", metrics=[])              # tables on the fused SGD update (lr 0.01)
    before = {k: v.weight.detach().clone() for k, v in lm.embedding_dict.items()}
    out = lm(X, sparse_feat_refine_weight=mx)
    assert tuple(out.shape) == (B, 1)
    assert max_abs(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-6
    out.backward(gy)
    torch.cuda.synchronize()
    assert max_abs(mx.grad.cpu().numpy(), mx_ref.grad.cpu().numpy()) <= 1e-6
    if dense:
        assert max_abs(lm.weight.grad.cpu().numpy(), w_ref.grad.cpu().numpy()) <= 1e-5
    lr = m.optim.param_groups[0]["lr"]
    for k, v in lm.embedding_dict.items():
        if v.weight.grad is not None:        # (tables outside the fused update: autograd's dense gradient)
            got = v.weight.grad
        else:
            got = (before[k] - v.weight.detach()) / lr
        gref = tabs[k].grad
        bar = 2e-5 * max(1.0, float(gref.abs().max())) + 2.0 ** -23 * float(before[k].abs().max()) / lr
        assert max_abs(got.cpu().numpy(), gref.cpu().numpy()) <= bar, k


def test_out_of_range_id_is_reported():
    g, m = _loaded("deepfm_fm_only")
    X = torch.from_numpy(g["X"].copy()).to(DEV)
    X[3, 0] = 1e6
    with torch.no_grad():
        m(X)
    with pytest.raises(IndexError):
        m.model_plan().check_ids()


def test_fit_predict_roundtrip_matches_reference_protocol():
    """fit() on device-resident data: History contents, predict() dtype/shape, loss goes down."""
    g, m = _loaded("deepfm_criteo")
    spec = g["spec"]
    names = [c["name"] for c in spec["dnn_columns"]]
    Xs = np.concatenate(list(g["extra"]["X_steps"]) + [g["X"]], axis=0)
    ys = np.concatenate(list(g["extra"]["y_steps"]) + [g["y"]], axis=0)
    x = {n: Xs[:, i] for i, n in enumerate(names)}
    m.compile("adagrad", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    hist = m.fit(x, ys, batch_size=64, epochs=3, verbose=2, validation_split=0.25, shuffle=True)
    assert set(hist.history) == {"loss", "binary_crossentropy", "auc", "val_binary_crossentropy", "val_auc"}
    assert len(hist.history["loss"]) == 3 and hist.history["loss"][-1] < hist.history["loss"][0]
    pred = m.predict(x, batch_size=50)
    assert pred.dtype == np.float64 and pred.shape == (Xs.shape[0], 1)
    assert np.all((pred > 0) & (pred < 1))
