"""CPU: every model family through the real Python stack over the stand-in for the library (tests/mock_lib.py +
tests/mock_ops.py), against the reference's golden forward values, per-parameter gradients and 3-step SGD / Adagrad
trajectories (tests/golden/*.npz), its own model-test matrix (tests/golden/matrix) and its fit() History.  Pins, without
a GPU, what sits between the reference-shaped API and the C-ABI: the plan's field / unit tables, buffer strides, the
choice of update mode, the dense-gradient route (param.grad), the in-kernel optimizer route and the marshalling of every
interaction op.  The kernels themselves are checked by tests/test_gpu_*.py."""
import numpy as np
import pytest
import torch

from helpers import build_model, golden_names, load_golden, max_abs

DEV = "cpu"
NAMES = golden_names()


def _loaded(name):
    g = load_golden(name)
    m = build_model(g["spec"], DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    return g, m


@pytest.mark.parametrize("name", NAMES)
def test_forward_matches_reference(mock, name):
    g, m = _loaded(name)
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g["X"]))
    assert max_abs(y.numpy(), g["y_pred"]) <= 2e-5
    m.model_plan().check_ids()


@pytest.mark.parametrize("name", NAMES)
def test_dense_gradients_match_reference(mock, name):
    g, m = _loaded(name)
    m.train()
    loss = torch.nn.functional.binary_cross_entropy(m(torch.from_numpy(g["X"])).squeeze(), torch.from_numpy(g["y"]),
                                                    reduction="sum")
    m.zero_grad()
    loss.backward()
    assert abs(loss.item() - g["loss"]) <= 1e-4 * max(1.0, abs(g["loss"]))
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        assert max_abs(got, ref) <= 2e-5 * max(1.0, float(np.max(np.abs(ref)))), k


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_in_kernel_optimizer_trajectory(mock, name, opt):
    g, m = _loaded(name)
    if (opt + "3_loss") not in g["extra"]:
        pytest.skip("no %s trajectory in this fixture" % opt)
    if opt == "adagrad" and name.startswith("afm"):
        pytest.skip("AFM under Adagrad is not a reproducible trajectory (sign of ~1e-8 gradients; see test_gpu_models)")
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    losses = [float(m._train_step(torch.from_numpy(Xb), torch.from_numpy(yb))[0])
              for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"])]
    plan = m.model_plan()
    if plan.table_params:
        # ("lazy", opt): DCNMix keeps the reference's quirk of regularising the linear tables with BaseModel's default
        # 1e-5 whatever l2_reg_linear says (dcnmix.py:52-54 does not forward it) -- the exact lazy form applies
        assert plan.update[0] in (opt, "sgd2", "lazy")
        if plan.unit_path and plan.update[0] == opt:
            assert "embed_update:%d" % (0 if opt == "sgd" else 1) in mock.calls
    np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=5e-5)
    sd = m.state_dict()
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            assert max_abs(sd[k[len(opt) + 2:]].numpy(), v) <= 1e-4, k


FIT_RUNS = (("plain", "adagrad", 0.0, False), ("shuffled", "adagrad", 0.0, True), ("default", "adam", 1e-5, True))


@pytest.mark.parametrize("tag,opt,l2,shuffle", FIT_RUNS)
def test_fit_history_and_predict_match_reference(mock, monkeypatch, tag, opt, l2, shuffle):
    """model.fit() of the REAL reference (tests/golden/fit_deepfm.npz: 3 epochs, batch 64 with a ragged last batch,
    validation split, per-batch metrics averaged over steps) -- same History, same predictions; with shuffle=True the
    same permutations as the reference's DataLoader draws after torch.manual_seed."""
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")           # no hipGraphs on CPU tensors
    g = load_golden("fit_deepfm")
    ex = g["extra"]
    m = build_model(g["spec"], DEV, l2=l2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.compile(opt, "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    x = {c["name"]: ex["fit_X"][:, i] for i, c in enumerate(g["spec"]["dnn_columns"])}
    torch.manual_seed(777)
    hist = m.fit(x, ex["fit_y"], batch_size=64, epochs=3, verbose=2, validation_split=0.25, shuffle=shuffle)
    ref = {k[len("fit_%s_hist/" % tag):]: v for k, v in ex.items() if k.startswith("fit_%s_hist/" % tag)}
    assert set(hist.history) == set(ref)
    for k, v in ref.items():
        np.testing.assert_allclose(hist.history[k], v, rtol=2e-4, err_msg=k)
    pred = m.predict(x, batch_size=50)
    assert pred.dtype == np.float64 and pred.shape == ex["fit_%s_pred" % tag].shape
    assert max_abs(pred, ex["fit_%s_pred" % tag]) <= 5e-5


# ---- the reference's own model-test matrix (tests/golden/matrix, oracle/check_matrix.py) on the stand-in ---------------
from helpers import feature_columns, load_matrix, matrix_id  # noqa: E402

def _matrix_model(c):
    import deepctr_torch.models as M
    spec = c["spec"]
    lin, dnn = feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"])
    cls = getattr(M, c["model"])
    m = cls(dnn, device=DEV, **c["kwargs"]) if c["model"] == "PNN" else cls(lin, dnn, device=DEV, **c["kwargs"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()})
    return m


@pytest.mark.parametrize("c", load_matrix(), ids=matrix_id)
def test_reference_matrix_forward_and_gradients(mock, c):
    """sum / mean / max VarLen columns (padding id 0 or a length column), one-row vocabularies, no-linear / no-FM /
    zero-layer-tower / empty-CIN / every bilinear and outer-product type ...: the plan's pooled-field descriptors, the
    general backward route and the argument marshalling of every interaction op (tests/mock_ops.py), for all ten model
    families, against the REAL reference's logits (1e-5) and per-parameter gradients."""
    m = _matrix_model(c)
    ok = c["clean"]
    m.eval()
    cap = {}
    hook = m.out.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("logit", inp[0].detach()))
    with torch.no_grad():
        m(torch.from_numpy(c["X"]))
    hook.remove()
    assert max_abs(cap["logit"].numpy().reshape(-1, 1)[ok], c["logit"][ok]) <= 1e-5
    m.train()
    okt = torch.from_numpy(ok)
    m.zero_grad()
    torch.nn.functional.binary_cross_entropy(m(torch.from_numpy(c["X"])).squeeze(1)[okt], torch.from_numpy(c["y"])[okt],
                                             reduction="sum").backward()
    for k, p in m.named_parameters():
        ref = c["grads"][k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        tol = 2e-5 * max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
        if k in c["grads64"]:
            # train-mode BatchNorm at initialisation is ill-conditioned (tests/test_gpu_reference_matrix.py): even torch-CPU
            # against itself differs with the thread count; bounded by the reference's own measured fp32 uncertainty
            tol = max(tol, 4.0 * max_abs(ref, c["grads64"][k]))
            ref = c["grads64"][k]
        assert max_abs(got, ref) <= tol, k


def test_reference_protocol_runs_on_pooled_fields(mock, monkeypatch, tmp_path):
    """check_model of the reference's tests (tests/utils.py:142-171) on a VarLen model: adam + default L2 ->
    the exact dense-gradient route ('dense' update mode), callbacks, state_dict and whole-model save / load."""
    from deepctr_torch.callbacks import EarlyStopping, ModelCheckpoint
    from deepctr_torch.models import DeepFM
    from matrix_data import N, make_data
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")
    x, y, cols = make_data(1, 3, 3)
    m = DeepFM(cols, cols, dnn_hidden_units=(32,), dnn_dropout=0.5, device=DEV)
    m.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "acc"])
    assert m.model_plan().update == ("dense",)
    ckpt = str(tmp_path / "model.ckpt")
    hist = m.fit(x, y, batch_size=100, epochs=2, validation_split=0.5, verbose=2, callbacks=[
        EarlyStopping(monitor="val_acc", min_delta=0, verbose=1, patience=1, mode="max"),
        ModelCheckpoint(filepath=ckpt, monitor="val_acc", verbose=1, save_best_only=True, save_weights_only=False,
                        mode="max", period=1)])
    assert set(hist.history) == {"loss", "binary_crossentropy", "acc", "val_binary_crossentropy", "val_acc"}
    # (round 5: pooled fields take the deterministic sorted update -- accumulate mode for a dense optimizer -- not the
    # float-atomic scatter)
    assert "embed_update:2" in mock.calls and not any(c.startswith("embed_bwd") for c in mock.calls)
    assert np.isfinite(hist.history["loss"]).all()
    w = str(tmp_path / "w.h5")
    torch.save(m.state_dict(), w)
    m.load_state_dict(torch.load(w))
    before = m.predict(x, batch_size=50)
    f = str(tmp_path / "m.h5")
    torch.save(m, f)
    again = torch.load(f, weights_only=False)
    assert before.shape == (N, 1) and max_abs(again.predict(x, batch_size=50), before) == 0.0


def test_history_longer_than_a_unit_stages_takes_the_two_pass_route(mock):
    """130 positions > DCTR_MAX_UNIT_SLOTS: the plan declines the unit path and the host routes the backward through
    dctr_embed_bwd (+ dctr_embed_apply under an in-kernel optimizer) -- gradients against autograd on the same formulas."""
    from deepctr_torch._hip.lib import MAX_UNIT_SLOTS
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_torch.models import DeepFM
    T = MAX_UNIT_SLOTS + 2
    cols = [SparseFeat("user", 50, 8), VarLenSparseFeat(SparseFeat("long_hist", 40, 8), maxlen=T, combiner="mean"),
            DenseFeat("price", 1)]
    torch.manual_seed(0)
    m = DeepFM(cols, cols, dnn_hidden_units=(16,), init_std=0.05, device=DEV)
    plan = m.model_plan()
    assert not plan.unit_path and plan.gen is None
    g = np.random.RandomState(1)
    B = 48
    n = g.randint(0, T + 1, (B, 1))
    seq = np.where(np.arange(T)[None, :] < n, g.randint(1, 40, (B, T)), 0)
    X = torch.from_numpy(np.concatenate([g.randint(0, 50, (B, 1)), seq, g.rand(B, 1)], axis=1).astype(np.float32))
    m.train()
    m(X).sum().backward()
    assert any(c.startswith("embed_bwd") for c in mock.calls) and not any(c.startswith("embed_update") for c in mock.calls)
    E = m.embedding_dict["long_hist"].weight
    got = E.grad.clone()
    # autograd over torch ops on the same parameters
    m.zero_grad()
    ids = X[:, 1:1 + T].long()
    mask = (ids != 0).float().unsqueeze(-1)
    pooled = (E[ids] * mask).sum(1) / (mask.sum(1) + 1e-8)
    Eu = m.embedding_dict["user"].weight[X[:, 0].long()]
    Wl = m.linear_model.embedding_dict
    wide = Wl["user"].weight[X[:, 0].long()] + ((Wl["long_hist"].weight[ids] * mask).sum(1) / (mask.sum(1) + 1e-8)) \
        + X[:, -1:] @ m.linear_model.weight
    s1 = Eu + pooled
    fm = 0.5 * (s1 * s1 - (Eu * Eu + pooled * pooled)).sum(1, keepdim=True)
    deep = m.dnn_linear(m.dnn(torch.cat([Eu, pooled, X[:, -1:]], dim=1)))
    torch.sigmoid(wide + fm + deep + m.out.bias).sum().backward()
    assert max_abs(got.numpy(), E.grad.numpy()) <= 2e-5 * max(1.0, float(E.grad.abs().max()))


def test_senet_beyond_the_kernel_envelope_matches_oracle(mock):
    """39 fields of 32 floats: more than dctr_senet_bwd stages in LDS -> the layer runs the reference's formulation as
    torch ops; values and gradients against the numpy oracle (pinned by the FiBiNET fixtures)."""
    from deepctr_torch.layers import SENETLayer
    from np_oracle import senet_backward, senet_forward
    torch.manual_seed(0)
    layer = SENETLayer(39, 3, device="cpu")
    E = torch.randn(6, 39, 32, requires_grad=True)
    V = layer(E)
    assert not any(c.startswith("senet") for c in mock.calls)
    W1, W2 = layer.excitation[0].weight, layer.excitation[2].weight
    want, cache = senet_forward(E.detach().double().numpy(), W1.detach().double().numpy(), W2.detach().double().numpy())
    assert max_abs(V.detach().numpy(), want) <= 1e-5
    gV = torch.randn(V.shape)
    gE, g1, g2 = torch.autograd.grad(V, [E, W1, W2], gV)
    oE, o1, o2 = senet_backward(gV.double().numpy(), E.detach().double().numpy(), cache, W1.detach().double().numpy(),
                                W2.detach().double().numpy())
    for got, ref in ((gE, oE), (g1, o1), (g2, o2)):
        assert max_abs(got.numpy(), ref) <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    small = SENETLayer(5, 2, device="cpu")
    small(torch.randn(4, 5, 8))
    assert "senet_fwd" in mock.calls            # inside the envelope the kernel path is taken


@pytest.mark.parametrize("btype", ["all", "each", "interaction"])
def test_bilinear_beyond_the_kernel_envelope_matches_the_kernel_formula(mock, btype):
    """39 fields of 16: the forward kernel would fit, the backward-data kernel's LDS tiles would not (ENOSUP after the
    forward) -> the layer must take the torch formulation for the whole op.  Checked against the stand-in of the kernel
    (tests/mock_ops.py: the formula include/dctr.h documents) on a shape inside the envelope, and for self-consistency of
    values / gradients beyond it."""
    from deepctr_torch.layers import BilinearInteraction
    assert BilinearInteraction._kernel_fits(37, 16) and not BilinearInteraction._kernel_fits(39, 16)
    assert not BilinearInteraction._kernel_fits(5, 20)
    torch.manual_seed(0)
    # same layer, same input: torch formulation vs kernel path (forced) on an in-envelope shape
    lay = BilinearInteraction(6, 8, btype, device="cpu")
    E = torch.randn(5, 6, 8, requires_grad=True)
    out_k = lay(E)
    assert "bilinear_fwd" in mock.calls
    out_t = lay._pairs_torch(E)
    assert max_abs(out_k.detach().numpy(), out_t.detach().numpy()) <= 1e-5
    g = torch.randn(out_k.shape)
    ws = [w for w in lay.parameters()]
    gk = torch.autograd.grad(out_k, [E] + ws, g, allow_unused=True)
    gt = torch.autograd.grad(out_t, [E] + ws, g, allow_unused=True)
    for a, b in zip(gk, gt):
        if a is None or b is None:          # 'each': the last field's weight is never a left factor
            assert (a is None or float(a.abs().max()) == 0) and (b is None or float(b.abs().max()) == 0)
        else:
            assert max_abs(a.numpy(), b.numpy()) <= 2e-5 * max(1.0, float(b.abs().max()))
    mock.calls.clear()
    big = BilinearInteraction(39, 16, btype, device="cpu")
    y = big(torch.randn(3, 39, 16))
    assert y.shape == (3, 39 * 38 // 2, 16) and not any(c.startswith("bilinear") for c in mock.calls)


def test_afm_beyond_the_backward_envelope_takes_the_torch_formulation(mock):
    """56 fields of 16 with attention_factor 8: the forward kernel fits, the backward's LDS image (every pair's attention
    and gradient rows) does not -> torch formulation for the whole op; equal to the kernel's formula on a small shape."""
    from deepctr_torch.layers import AFMLayer
    assert AFMLayer._kernel_fits(39, 16, 8) and not AFMLayer._kernel_fits(56, 16, 8) and not AFMLayer._kernel_fits(30, 64, 32)
    torch.manual_seed(0)
    lay = AFMLayer(8, 4, device="cpu")
    E = torch.randn(5, 6, 8, requires_grad=True)
    y_k = lay(E)
    assert "afm_fwd" in mock.calls
    lay._kernel_fits = staticmethod(lambda F, D, A: False)
    y_t = lay(E)
    assert max_abs(y_k.detach().numpy(), y_t.detach().numpy()) <= 1e-5
    ps = [E] + list(lay.parameters())
    g = torch.randn(y_k.shape)
    for a, b in zip(torch.autograd.grad(y_k, ps, g), torch.autograd.grad(y_t, ps, g)):
        assert max_abs(a.numpy(), b.numpy()) <= 2e-5 * max(1.0, float(b.abs().max()))
    mock.calls.clear()
    big = AFMLayer(16, 8, device="cpu")
    assert big(torch.randn(3, 56, 16)).shape == (3, 1) and "afm_fwd" not in mock.calls


def test_crossnet_beyond_the_backward_envelope_takes_the_torch_formulation(mock):
    """6 layers over 2000 inputs: the forward kernel fits, the backward's LDS image (4 x layers x W floats) does not."""
    from deepctr_torch.layers import CrossNet
    torch.manual_seed(0)
    small = CrossNet(12, 3, "vector", device="cpu")
    with torch.no_grad():
        small.bias.normal_(0, 0.1)
    x = torch.randn(5, 12, requires_grad=True)
    y_k = small(x)
    assert "crossnet_vec_fwd" in mock.calls
    x_0 = x_l = x
    for i in range(3):
        x_l = x_0 * torch.matmul(x_l, small.kernels[i]) + small.bias[i].squeeze(1) + x_l
    assert max_abs(y_k.detach().numpy(), x_l.detach().numpy()) <= 1e-5
    mock.calls.clear()
    big = CrossNet(2000, 6, "vector", device="cpu")
    assert big(torch.randn(2, 2000)).shape == (2, 2000) and "crossnet_vec_fwd" not in mock.calls
    ok = CrossNet(429, 6, "vector", device="cpu")          # the Criteo width: inside
    ok(torch.randn(2, 429))
    assert "crossnet_vec_fwd" in mock.calls


def test_inner_products_beyond_the_backward_envelope(mock):
    """45 fields of 16 without the sum over d: one sample's P gradient rows no longer fit the backward kernel's LDS."""
    from deepctr_torch.layers import InnerProductLayer
    from deepctr_torch.layers.interaction import pairwise_products
    torch.manual_seed(0)
    E = torch.randn(4, 6, 8, requires_grad=True)
    for reduce in (True, False):
        y_k = pairwise_products(E, reduce)
        idx = torch.triu_indices(6, 6, 1)
        y_t = E[:, idx[0]] * E[:, idx[1]]
        y_t = y_t.sum(dim=2, keepdim=True) if reduce else y_t
        assert y_k.shape == y_t.shape and max_abs(y_k.detach().numpy(), y_t.detach().numpy()) <= 1e-6
    assert mock.calls.count("inner_product_fwd") == 2
    mock.calls.clear()
    big = torch.randn(2, 45, 16)
    assert pairwise_products(big, False).shape == (2, 990, 16) and "inner_product_fwd" not in mock.calls
    assert InnerProductLayer(device="cpu")([t for t in big.split(1, dim=1)]).shape == (2, 990, 1)
    assert "inner_product_fwd" in mock.calls            # with the sum the backward image is small: kernel


@pytest.mark.parametrize("name", ["deepfm_criteo", "dcn_vector"])
def test_forward_hooks_keep_firing_in_train_steps(mock, name):
    """The fast train steps do not call ``model.out`` / ``model.forward`` (fused step: one tower + head + loss kernel;
    autograd route: one head kernel).  A user's forward hook on either must still fire, as under the reference
    (basemodel.py:242-254 calls ``model(x)``): with hooks registered the step takes the stock module route, and
    lands on the same parameters."""
    g = load_golden(name)
    finals = []
    for hooked in (False, True):
        m = build_model(g["spec"], "cpu")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        seen = []
        if hooked:
            m.out.register_forward_pre_hook(lambda mod, inp: seen.append(tuple(inp[0].shape)))
        mock.calls.clear()
        for _ in range(2):
            m._train_step(torch.from_numpy(g["X"]), torch.from_numpy(g["y"]))
        assert len(seen) == (2 if hooked else 0)
        fast = any(c in ("mlp_train_step", "bce_head", "embed_tower_train_step") for c in mock.calls)
        assert fast != hooked
        finals.append({k: v.clone() for k, v in m.state_dict().items()})
    for k in finals[0]:
        assert max_abs(finals[0][k].numpy(), finals[1][k].numpy()) <= 2e-5 * max(1.0, float(finals[0][k].abs().max())), k


def test_weights_only_checkpoint_is_the_references_size(mock, tmp_path):
    """After compile('adagrad') a table is a strided view of a slab that interleaves it with its Adagrad state
    (_hip/layout.py).  state_dict() -- and optimizer.state_dict() -- must still hand out plain contiguous tensors: a saved
    view drags its whole storage along (2x the table's bytes, accumulators included: round-2 advisor finding)."""
    import io
    g, m = _loaded("deepfm_criteo")
    plain = io.BytesIO()
    torch.save(m.state_dict(), plain)
    m.compile("adagrad", "binary_crossentropy", metrics=[])
    m.train()
    m._train_step(torch.from_numpy(g["extra"]["X_steps"][0]), torch.from_numpy(g["extra"]["y_steps"][0]))
    w = m.embedding_dict["C1"].weight
    assert not w.is_contiguous(), "the test needs the interleaved layout"
    sd = m.state_dict()
    assert all(v.is_contiguous() for v in sd.values())
    assert torch.equal(sd["embedding_dict.C1.weight"], w.detach())
    after = io.BytesIO()
    torch.save(sd, after)
    assert after.getbuffer().nbytes <= plain.getbuffer().nbytes * 1.02
    osd = m.optim.state_dict()
    assert all(v.is_contiguous() for st in osd["state"].values() for v in st.values() if torch.is_tensor(v))
    # and both still load back
    m.load_state_dict(sd)
    m.optim.load_state_dict(osd)


def test_step_engine_takes_pooled_sum_and_mean_fields(mock, monkeypatch):
    """Round 5: a DeepFM with a mean history (ids != 0 mask) and a sum history over a SHARED table (length column) runs on the
    step engine (deepctr_torch/_hip/step.py: general update units, per-step den_t / amax buffers, n_vcols-row id arrays) and
    lands where the autograd-assembled step lands; so does one with a max-pooled column."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_torch.models import DeepFM

    def cols(with_max):
        c = [SparseFeat("a", 30, 8), SparseFeat("b", 20, 8), DenseFeat("x", 2),
             VarLenSparseFeat(SparseFeat("hist", 25, 8), maxlen=4, combiner="mean"),
             VarLenSparseFeat(SparseFeat("seq", 30, 8, embedding_name="a"), maxlen=3, combiner="sum", length_name="seq_len")]
        if with_max:
            c.append(VarLenSparseFeat(SparseFeat("kw", 9, 8), maxlen=2, combiner="max"))
        return c

    rng = np.random.RandomState(0)
    B = 48

    def data(with_max):
        h = rng.randint(1, 25, (B, 4)) * (np.arange(4)[None, :] < rng.randint(0, 5, (B, 1)))
        parts = [rng.randint(0, 30, (B, 1)), rng.randint(0, 20, (B, 1)), rng.rand(B, 2), h, rng.randint(0, 30, (B, 3)),
                 rng.randint(0, 4, (B, 1))]
        if with_max:
            parts.append(rng.randint(1, 9, (B, 2)))
        return torch.from_numpy(np.concatenate(parts, axis=1).astype(np.float32)), \
            torch.from_numpy(rng.randint(0, 2, B).astype(np.float32))

    def run(engine, with_max, X, y):
        monkeypatch.setenv("DCTR_STEP_ENGINE", "1" if engine else "0")
        torch.manual_seed(0)
        c = cols(with_max)
        m = DeepFM(c, c, dnn_hidden_units=(16, 8), l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0, device=DEV)
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        mock.calls.clear()
        losses = [float(m._train_step(X, y)[0]) for _ in range(3)]
        st = m._fused_step_state()
        used = st is not None and st.get("engine") is not None and st["engine"].steps_run > 0
        return m, losses, used, list(mock.calls)

    X, y = data(False)
    m1, l1, used1, calls1 = run(True, False, X, y)
    m0, l0, used0, calls0 = run(False, False, X, y)
    assert used1 and not used0 and "embed_tower_train_step" in calls1 and "embed_tower_train_step" not in calls0
    assert not m1.model_plan().simple_units and m1.model_plan().gen is not None
    assert not any(c.startswith("embed_bwd") for c in calls1 + calls0)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m0.state_dict().items()):
        assert max_abs(a.numpy(), b.numpy()) <= 1e-6, k
    # a max-pooled column too: its arg-max positions are a side output of the tower launch's gather stage
    Xm, ym = data(True)
    mm1, lm1, usedm1, callsm1 = run(True, True, Xm, ym)
    mm0, lm0, usedm0, _ = run(False, True, Xm, ym)
    assert usedm1 and not usedm0 and "embed_update:1" in callsm1 and "embed_tower_train_step" in callsm1
    np.testing.assert_allclose(lm1, lm0, rtol=1e-6)
    for (k, a), (_, b) in zip(mm1.state_dict().items(), mm0.state_dict().items()):
        assert max_abs(a.numpy(), b.numpy()) <= 1e-6, k


def test_forward_only_layout_round_trip_on_the_stand_in(mock):
    """_hip/layout.py apply_infer_layout (round 5): predict() on a never-compiled model seats deep row + wide weight of an id
    in one [V, 32] slab; values, state_dict and a later compile('adagrad') (re-seating into the interleaved slabs -- which must
    NOT adopt the forward-only slab of the same width: the Adagrad state would land on the wide weights) are unaffected."""
    g = load_golden("deepfm_criteo")
    m = build_model(g["spec"], DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    X = g["X"]
    x = {name: (X[:, lo] if hi - lo == 1 else X[:, lo:hi]) for name, (lo, hi) in m.feature_index.items()}
    pred = m.predict(x, batch_size=64)
    plan = m.model_plan()
    n = 0
    for di, wi, _, _ in plan.units:
        if di >= 0 and wi >= 0 and plan.deep[di].dim % 4 == 0 and plan.deep[di].dim <= 28:
            pd, pw = plan.deep[di].param, plan.wide[wi].param
            assert pd.stride(0) == 32 and pw.stride(0) == 32 and pw.data_ptr() == pd.data_ptr() + 4 * plan.deep[di].dim
            n += 1
    assert n > 0
    assert max_abs(pred.reshape(-1), g["y_pred"].reshape(-1)) <= 1e-5
    sd = m.state_dict()
    for k, v in g["params"].items():
        assert sd[k].is_contiguous() and max_abs(sd[k].numpy(), v) == 0.0, k
    m.compile("adagrad", "binary_crossentropy", metrics=[])
    for di, wi, _, _ in plan.units:
        if wi >= 0:
            assert plan.wide[wi].param.stride(0) == 2           # [V, 2]: weight | Adagrad sum
    sd2 = m.state_dict()
    for k, v in g["params"].items():
        assert max_abs(sd2[k].numpy(), v) == 0.0, "%s changed while re-seating" % k
    m.train()
    losses = [float(m._train_step(torch.from_numpy(Xb), torch.from_numpy(yb))[0])
              for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"])]
    np.testing.assert_allclose(losses, g["extra"]["adagrad3_loss"], rtol=5e-5)
