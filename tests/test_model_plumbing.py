"""CPU: DeepFM / WDL through the real Python stack over the numpy stand-in for the library (tests/mock_lib.py), against
the reference's golden forward values, per-parameter gradients and 3-step SGD / Adagrad trajectories
(tests/golden/*.npz).  Pins, without a GPU, what sits between the reference-shaped API and the C-ABI: the plan's field /
unit tables, buffer strides, the choice of update mode, the dense-gradient route (param.grad) and the in-kernel
optimizer route.  Fixed-length fields only (the stand-in's scope); the kernels are checked by tests/test_gpu_*.py."""
import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs

DEV = "cpu"
NAMES = ["deepfm_criteo", "deepfm_dense_only", "deepfm_fm_only", "wdl_criteo"]


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
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    losses = [float(m._train_step(torch.from_numpy(Xb), torch.from_numpy(yb))[0])
              for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"])]
    if m.model_plan().table_params:
        assert m.model_plan().update[0] == opt
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


# ---- the reference's own DeepFM test matrix (tests/golden/matrix, oracle/check_matrix.py) on the stand-in --------------
from helpers import feature_columns, load_matrix, matrix_id  # noqa: E402

DEEPFM_MATRIX = [c for c in load_matrix() if c["model"] == "DeepFM"]


def _matrix_model(c):
    from deepctr_torch.models import DeepFM
    spec = c["spec"]
    m = DeepFM(feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"]), device=DEV, **c["kwargs"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()})
    return m


@pytest.mark.parametrize("c", DEEPFM_MATRIX, ids=matrix_id)
def test_reference_matrix_forward_and_gradients(mock, c):
    """sum / mean / max VarLen columns (padding id 0 or a length column), one-row vocabularies, no-linear / no-FM /
    zero-layer-tower variants: the plan's pooled-field descriptors and the general backward route, against the REAL
    reference's logits (1e-5) and per-parameter gradients."""
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
        assert max_abs(got, ref) <= 2e-5 * max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0), k


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
    assert "embed_bwd:0" in mock.calls and np.isfinite(hist.history["loss"]).all()
    w = str(tmp_path / "w.h5")
    torch.save(m.state_dict(), w)
    m.load_state_dict(torch.load(w))
    before = m.predict(x, batch_size=50)
    f = str(tmp_path / "m.h5")
    torch.save(m, f)
    again = torch.load(f, weights_only=False)
    assert before.shape == (N, 1) and max_abs(again.predict(x, batch_size=50), before) == 0.0
