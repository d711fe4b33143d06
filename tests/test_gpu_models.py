"""GPU parity of the other drop-in model families (xDeepFM / FiBiNET / DCN / PNN) against the golden vectors the
real reference produced (tests/golden, oracle/make_golden.py) and the numpy oracle: pre-sigmoid logits within 1e-5
(north_star), every parameter gradient, and 3-step SGD / Adagrad trajectories with the fused sparse update."""
import numpy as np
import pytest
import torch

from helpers import build_model, golden_names, load_golden, max_abs
from np_oracle import Oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL, GRAD_TOL, TRAJ_TOL = 1e-5, 2e-5, 2e-5
NAMES = [n for n in golden_names() if not n.startswith("deepfm")]


def _loaded(name, l2=0.0):
    g = load_golden(name)
    m = build_model(g["spec"], DEV, l2=l2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    return g, m


@pytest.mark.parametrize("name", NAMES)
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
    err = max_abs(cap["logit"].cpu().numpy(), g["logit"])
    assert err <= LOGIT_TOL, "logit max|d|=%.3e" % err
    assert max_abs(y.cpu().numpy(), g["y_pred"]) <= LOGIT_TOL
    l64, _ = Oracle(g["spec"], g["params"], dtype=np.float64).forward(g["X"])
    assert max_abs(cap["logit"].cpu().numpy(), l64) <= LOGIT_TOL


@pytest.mark.parametrize("name", NAMES)
def test_gradients_match_reference(name):
    g, m = _loaded(name)
    m.train()
    X, y = torch.from_numpy(g["X"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    loss = torch.nn.functional.binary_cross_entropy(m(X).squeeze(), y, reduction="sum")
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - g["loss"]) <= 1e-4 * max(1.0, abs(g["loss"]))
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        scale = max(1.0, float(np.max(np.abs(ref))))
        err = max_abs(got, ref)
        assert err <= GRAD_TOL * scale, "%s: max|d|=%.3e scale %.3g" % (k, err, scale)


@pytest.mark.parametrize("name", [n for n in NAMES if "X_steps" in load_golden(n)["extra"]])
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_training_trajectory_matches_reference(name, opt):
    g, m = _loaded(name)
    if (opt + "3_loss") not in g["extra"]:
        pytest.skip("no %s trajectory in this fixture" % opt)
    if opt == "adagrad" and name.startswith("afm"):
        # From ZERO accumulators Adagrad's first steps are lr * sign(g), and AFM's attention path yields gradients that
        # cancel to ~1e-8 whose sign is rounding noise: three implementations of these three steps (the reference in fp32,
        # the numpy oracle in fp64 and in fp32) differ by 1e-2 on attention_b.  AFM's Adagrad trajectory is therefore
        # pinned from PRESET accumulators instead (round 6: test_adagrad_trajectory_from_preset_accumulators below,
        # every element against the reference alone); this zero-start variant has no defined answer for AFM.
        return
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    losses = []
    for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
        loss, _, _ = m._train_step(torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV))
        losses.append(loss.item())
    m.model_plan().check_ids()
    np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=2e-5)
    sd = m.state_dict()
    # Under Adagrad the first steps are lr * sign(g)-like, so an element whose gradient cancels to ~1e-8 (AFM's
    # attention path produces such elements) lands on either side depending on fp32 rounding: there the reference's own
    # fp32 run sits up to 2.5e-3 away from an fp64 evaluation of the same three steps.  Every element must agree with
    # the reference OR with the fp64 oracle trajectory (oracle/np_oracle.py); per-step gradients and the SGD
    # trajectories are checked against the reference alone.
    o64 = None
    if opt == "adagrad":
        o64 = Oracle(g["spec"], g["params"], dtype=np.float64)
        st64 = None
        for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
            _, st64 = o64.train_step(Xb, yb, optimizer="adagrad", lr=0.01, eps=1e-10, state=st64)
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            key = k[len(opt) + 2:]
            got = sd[key].cpu().numpy().astype(np.float64)
            d = np.abs(got - v.astype(np.float64))
            if o64 is not None:
                d = np.minimum(d, np.abs(got - np.asarray(o64.P[key], np.float64).reshape(got.shape)))
            err = float(d.max()) if d.size else 0.0
            assert err <= TRAJ_TOL, "%s: %.3e" % (key, err)


@pytest.mark.parametrize("name", [n for n in golden_names() if "adagradp3_loss" in load_golden(n)["extra"]])
def test_adagrad_trajectory_from_preset_accumulators(name):
    """``adagradp3`` (oracle/make_golden.py): the reference's three Adagrad steps with every accumulator preset to 0.05 -- the
    step lr * g / sqrt(s0 + g^2) is smooth in g, so EVERY element of every parameter must land on the reference's fp32
    result within TRAJ_TOL: no fp64 alternative, no tolerated fraction, AFM included."""
    g, m = _loaded(name)
    m.compile("adagrad", "binary_crossentropy", metrics=[])
    for grp in m.optim.param_groups:
        for p in grp["params"]:
            m.optim.state[p]["sum"].fill_(0.05)
    m.train()
    losses = []
    for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
        loss, _, _ = m._train_step(torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV))
        losses.append(loss.item())
    torch.cuda.synchronize()
    m.model_plan().check_ids()
    np.testing.assert_allclose(losses, g["extra"]["adagradp3_loss"], rtol=2e-5)
    sd = m.state_dict()
    for k, v in g["extra"].items():
        if k.startswith("adagradp3/"):
            err = max_abs(sd[k[10:]].cpu().numpy(), v)
            assert err <= TRAJ_TOL, "%s: %.3e" % (k, err)


@pytest.mark.parametrize("name", ["dcn_vector", "xdeepfm_criteo", "afm_criteo"])
def test_fit_replays_graphs_for_autograd_step_models(name, monkeypatch):
    """fit() also replays a hipGraph of the AUTOGRAD train step (models outside the fused step) when that is
    replay-safe: same parameters and History as with graphs switched off."""
    g = load_golden(name)
    names = []
    for c in g["spec"]["linear_columns"] + g["spec"]["dnn_columns"]:
        if c["name"] not in names:
            names.append(c["name"])
    n = (g["X"].shape[0] // 16) * 16
    X = np.concatenate([g["X"][:n]] * 3, axis=0)
    y = np.concatenate([g["y"][:n]] * 3, axis=0)
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("DCTR_FIT_GRAPH", flag)
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        fi = m.feature_index
        x = {nm: X[:, fi[nm][0]] for nm in names}
        hist = m.fit(x, y, batch_size=16, epochs=2, verbose=0, shuffle=False)
        used = m._fit_graph is not None and m._fit_graph.get("graph") is not None
        runs.append(({k: v.clone() for k, v in m.state_dict().items()}, dict(hist.history), used))
    (a, ha, ua), (b, hb, ub) = runs
    assert ua and not ub
    for k in a:
        err = max_abs(a[k].cpu().numpy(), b[k].cpu().numpy())
        assert err <= 1e-6 * max(1.0, float(b[k].abs().max())), "%s: %.3e" % (k, err)
    np.testing.assert_allclose(ha["loss"], hb["loss"], rtol=1e-6)
