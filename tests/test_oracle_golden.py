"""Pin the numpy oracle (oracle/np_oracle.py) to the fixtures produced by the real reference.

CPU only.  Forward: logits / predictions / loss; backward: the gradient of EVERY parameter (dense [V, D]
embedding gradients included); trajectories: 3 reference training steps under SGD and Adagrad."""
import os

import numpy as np
import pytest

from helpers import golden_names, load_golden, max_abs
from np_oracle import Oracle, bce_sum

LOGIT_TOL = 1e-5   # north_star: logits within 1e-5 of the reference's fp32-CPU logits
GRAD_TOL = 2e-5    # fp32 re-association noise on sums over the batch


@pytest.mark.parametrize("name", golden_names())
def test_forward_matches_reference(name):
    g = load_golden(name)
    o = Oracle(g["spec"], g["params"])
    logit, y_pred = o.forward(g["X"])
    assert max_abs(logit, g["logit"]) <= LOGIT_TOL
    assert max_abs(y_pred, g["y_pred"]) <= LOGIT_TOL
    loss = bce_sum(y_pred.astype(np.float64), g["y"].reshape(-1, 1).astype(np.float64))
    assert abs(loss - g["loss"]) <= 1e-4 * max(1.0, abs(g["loss"]))


@pytest.mark.parametrize("name", golden_names())
def test_backward_matches_reference(name):
    g = load_golden(name)
    # fp64: numpy's fp32 einsum accumulates naively and is noisier than the reference itself (whose fp32
    # gradients sit within ~3e-6 relative of this fp64 evaluation)
    o = Oracle(g["spec"], g["params"], dtype=np.float64)
    _, y_pred = o.forward(g["X"])
    grads = o.backward(y_pred - g["y"].reshape(-1, 1))       # d BCE(sum) / d logit
    assert set(g["grads"]) <= set(grads) | {k for k, v in g["grads"].items() if not np.any(v)}
    for k, ref in g["grads"].items():
        got = np.asarray(grads.get(k, np.zeros_like(ref))).reshape(ref.shape)
        scale = max(1.0, float(np.max(np.abs(ref))))
        assert max_abs(got, ref) <= GRAD_TOL * scale, k


@pytest.mark.parametrize("name", [n for n in golden_names("deepfm") if "X_steps" in load_golden(n)["extra"]])
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_training_trajectory_matches_reference(name, opt):
    g = load_golden(name)
    o = Oracle(g["spec"], g["params"])
    state = None
    losses = []
    for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
        loss, state = o.train_step(Xb, yb, optimizer=opt, lr=0.01, eps=1e-10, state=state)
        losses.append(loss)
    np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=2e-5)
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            key = k[len(opt) + 2:]
            assert max_abs(o.P[key], v) <= 2e-5, key


def test_fp64_oracle_brackets_fp32_noise():
    """The fp32 reference is within ~1e-6 of an fp64 evaluation at |logit| ~ 4."""
    g = load_golden("deepfm_criteo")
    l64, _ = Oracle(g["spec"], g["params"], dtype=np.float64).forward(g["X"])
    assert max_abs(l64, g["logit"]) <= 5e-6


def test_torch_port_matches_reference():
    """The torch-CPU restatement used as bench.py's cpu_baseline reproduces the reference's logits and
    3-step Adagrad trajectory on the Criteo-shaped fixture."""
    import torch
    from torch_port import DeepFMPort, make_optimizer, train_step
    g = load_golden("deepfm_criteo")
    cols = g["spec"]["dnn_columns"]
    sparse = [c for c in cols if c["kind"] == "sparse"]
    port = DeepFMPort(len(sparse), [c["vocab"] for c in sparse], sparse[0]["dim"],
                      len(cols) - len(sparse), hidden=g["spec"]["kwargs"]["dnn_hidden_units"])
    port.load_reference_state(g["params"], [c["name"] for c in sparse])
    with torch.no_grad():
        logit = port.logit(torch.from_numpy(g["X"]))
    assert max_abs(logit.numpy(), g["logit"]) <= LOGIT_TOL
    opt = make_optimizer(port, "adagrad")
    losses = [train_step(port, opt, torch.from_numpy(Xb), torch.from_numpy(yb)).item()
              for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"])]
    np.testing.assert_allclose(losses, g["extra"]["adagrad3_loss"], rtol=2e-5)
    assert max_abs(port.emb[0].weight.detach().numpy(), g["extra"]["adagrad3/embedding_dict.C1.weight"]) <= 2e-5


STEP_FIXTURES = [n for n in golden_names() if "adagradp3_loss" in np.load(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", n + ".npz")).files]


@pytest.mark.parametrize("name", STEP_FIXTURES)
def test_oracle_adagrad_trajectory_from_preset_accumulators(name):
    """``adagradp3``: three reference Adagrad steps with every accumulator preset to 0.05 (oracle/make_golden.py) -- the step is
    then smooth in the gradient (no lr * sign(g) first step), so the fp64 oracle must land on the reference's fp32
    parameters on EVERY element, AFM's attention path included."""
    g = load_golden(name)
    o = Oracle(g["spec"], g["params"], dtype=np.float64)
    st = {k: np.full(np.shape(v), 0.05, np.float64) for k, v in o.P.items()}
    losses = []
    for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
        lo, st = o.train_step(Xb, yb, optimizer="adagrad", lr=0.01, eps=1e-10, state=st)
        losses.append(lo)
    np.testing.assert_allclose(losses, g["extra"]["adagradp3_loss"], rtol=2e-5)
    for k, v in g["extra"].items():
        if k.startswith("adagradp3/"):
            assert max_abs(o.P[k[10:]], v) <= 2e-5, k


# ---- the reference's own model-test matrix (tests/golden/matrix, oracle/check_matrix.py) -----------------------------
from helpers import feature_columns, load_matrix, matrix_id  # noqa: E402

MATRIX = load_matrix()
ORACLE_MATRIX = [c for c in MATRIX if c.get("oracle", True)]      # (PReLU / BatchNorm / sigmoid towers: reference only)


@pytest.mark.parametrize("c", ORACLE_MATRIX, ids=matrix_id)
def test_oracle_matches_reference_on_its_test_matrix(c):
    """Every configuration of tests/models/*_test.py of the reference (1-9 row vocabularies, sum / mean / max VarLen
    columns with padding id 0, no-linear / no-FM / empty-tower / empty-CIN variants ...): oracle == reference forward on
    the reference's own freshly initialised parameters."""
    logit, y_pred = Oracle(c["spec"], c["params"], dtype=np.float64).forward(c["X"])
    ok = c["clean"]
    assert ok.sum() >= 16
    assert max_abs(np.asarray(logit).reshape(-1, 1)[ok], c["logit"][ok]) <= LOGIT_TOL
    assert max_abs(np.asarray(y_pred)[ok], c["y_pred"][ok]) <= 2e-6


@pytest.mark.parametrize("c", ORACLE_MATRIX, ids=matrix_id)
def test_oracle_backward_matches_reference_on_its_test_matrix(c):
    """Gradient of BCE(sum) over the rows with a defined value, every parameter (a parameter the model never uses, e.g.
    the last of FiBiNET's 'each' bilinear weights, has the gradient zero)."""
    o = Oracle(c["spec"], c["params"], dtype=np.float64)
    _, y_pred = o.forward(c["X"])
    g = o.backward((np.asarray(y_pred).reshape(-1, 1) - c["y"].reshape(-1, 1).astype(np.float64)) * c["clean"].reshape(-1, 1))
    for k, ref in c["grads"].items():
        got = np.asarray(g.get(k, np.zeros_like(ref))).reshape(ref.shape)
        assert max_abs(got, ref) <= GRAD_TOL * max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0), k


@pytest.mark.parametrize("c", MATRIX, ids=matrix_id)
def test_state_dict_layout_matches_reference_on_its_test_matrix(c):
    """Drop-in checkpoints: same keys, same shapes as the reference's ``state_dict()`` for every configuration."""
    import deepctr_torch.models as M
    spec = c["spec"]
    lin, dnn = feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"])
    cls = getattr(M, c["model"])
    m = cls(dnn, device="cpu", **c["kwargs"]) if c["model"] == "PNN" else cls(lin, dnn, device="cpu", **c["kwargs"])
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v.shape) for k, v in c["params"].items()}
    assert list(m.state_dict().keys()) == list(c["params"].keys())          # same registration order
    # ... and the same VALUES: construction consumes the seeded generator like the reference (same modules, same order,
    # same init functions), so seed=1024 yields the reference's own initial weights, bit for bit
    for k, v in c["params"].items():
        assert np.array_equal(m.state_dict()[k].numpy(), v), k


def test_full_size_fixture_inputs_regenerate_bit_for_bit():
    """tests/golden/full/*.npz (the real reference at 26 x 1M x 16, batch 4096: oracle/make_full_golden.py) store outputs
    only; inputs and parameters come from tests/fullsize_data.py's integer hash on both sides.  The generator must give here
    what it gave in the run that produced the fixtures."""
    import fullsize_data as FD
    for name in FD.MODELS:
        X, y = FD.inputs(FD.data_of(name))
        touched = FD.touched_rows(X, FD.data_of(name))
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full", "%s.npz" % name))
        assert float(z["check/X"]) == float(X.astype(np.float64).sum())
        assert float(z["check/y"]) == float(y.sum())
        assert float(z["check/table_C7"]) == float(
            FD.table_rows("embedding_dict.C7.weight", touched[6], FD.DIM).astype(np.float64).sum())
        assert float(z["check/dnn0"]) == float(FD.dense_param("dnn.linears.0.weight", (8, 429)).astype(np.float64).sum())
        assert np.array_equal(z["n_touched"], np.array([len(r) for r in touched]))
        assert z["logit"].shape == (FD.BATCH,) and np.isfinite(z["logit"]).all()
        assert 1.0 < float(np.abs(z["logit"]).max()) < 10.0          # trained-like scale: the 1e-5 bar means something
