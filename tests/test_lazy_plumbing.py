"""CPU: the whole Python stack of a train step (EmbeddingPlan, EmbedFunction, LazyState, DenseSlab, the fused step of
BaseModel) over the numpy stand-in for the library (tests/mock_lib.py), replaying the REAL reference's 8-step
trajectories of tests/golden/lazy_*.npz (oracle/make_golden.py): L2 on every table under SGD / Adagrad / Adam, and
Adam without L2.  What this pins without a GPU: the call protocol (ids -> catch-up -> gather -> update(ACCUM) -> apply ->
step_inc, flush before any other reader), the ctypes marshalling of dctr_plan_t / dctr_lazy_unit_t, the lambda / state
wiring per table, and the logged loss.  The kernels' own arithmetic is covered by tests/test_gpu_lazy.py."""
import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs

DEV = "cpu"
TAGS = ("sgd", "adagrad", "adam", "adam0")


def _close(tag, got, ref, tol=5e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
    err = max_abs(got, ref)
    assert err <= tol * scale, "%s: max|d| = %.3e (scale %.3e)" % (tag, err, scale)


def _run(name, tag, predict_at=()):
    g = load_golden(name)
    ex = g["extra"]
    m = build_model(g["spec"], DEV, l2=0.0 if tag == "adam0" else 1e-3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.compile("adam" if tag == "adam0" else tag, "binary_crossentropy", metrics=[])
    m.train()
    assert m.model_plan().update == ("lazy", "adam" if tag == "adam0" else tag)
    bce, tot = [], []
    for i in range(len(ex["lazy_X"])):
        m.train()
        loss, total, _ = m._train_step(torch.from_numpy(ex["lazy_X"][i]), torch.from_numpy(ex["lazy_y"][i]))
        bce.append(float(loss))
        tot.append(float(total))
        if i in predict_at:
            m.eval()
            with torch.no_grad():
                m(torch.from_numpy(ex["lazy_X"][0]))
    return g, m, bce, tot


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", ["lazy_deepfm", "lazy_dcn"])
def test_lazy_stack_replays_reference_trajectory(mock, name, tag):
    g, m, bce, tot = _run(name, tag)
    ex = g["extra"]
    if name == "lazy_deepfm":
        # one fused step = ids, catch-up, the sweep of one K-th of the rows (beside the rest of the step on the GPU), gather (+ the segment pre-pass of the update,
        # right behind it), tower step, accumulate, apply, dense optimizer -- in this order
        first = mock.calls[:mock.calls.index("dense_opt_reg") + 1]
        assert first == ["embed_ids", "lazy_catchup", "lazy_sweep", "embed_fwd", "embed_segments", "mlp_train_step",
                         "embed_update:2", "lazy_apply", "dense_opt_reg"], first
    else:
        # DCN is outside the fused step: autograd + torch.optim around the kernels, the tables still lazily
        i0 = mock.calls.index("lazy_apply")
        assert mock.calls[:4] == ["embed_ids", "lazy_catchup", "lazy_sweep", "embed_fwd"] and \
            "embed_update:2" in mock.calls[:i0]
    np.testing.assert_allclose(bce, ex["lazy_%s_bce" % tag], rtol=5e-5)
    np.testing.assert_allclose(tot, ex["lazy_%s_total" % tag], rtol=5e-5)
    n_flush = mock.calls.count("lazy_flush")
    sd = m.state_dict()                                   # must flush: every row replayed to the current step
    assert mock.calls.count("lazy_flush") == n_flush + 1
    for k, v in ex.items():
        if k.startswith("lazy_%s/" % tag):
            _close(k, sd[k[len("lazy_%s/" % tag):]].numpy(), v)
    m.eval()
    with torch.no_grad():
        pred = m(torch.from_numpy(ex["lazy_X"][0]))
    _close("pred", pred.numpy().reshape(-1, 1), ex["lazy_%s_pred" % tag])
    p0 = m.embedding_dict[g["spec"]["dnn_columns"][0]["embedding_name"]].weight
    st = m.optim.state[p0]
    for key in ("sum", "exp_avg", "exp_avg_sq"):
        ref = ex.get("lazy_%s_state_%s" % (tag, key))
        if ref is not None:
            _close("state." + key, st[key].numpy(), ref)
    if tag.startswith("adam"):
        assert float(st["step"]) == len(ex["lazy_X"])


def test_flush_between_steps_keeps_the_trajectory(mock):
    g, m, _, _ = _run("lazy_deepfm", "adam", predict_at=(2, 5))
    sd = m.state_dict()
    for k, v in g["extra"].items():
        if k.startswith("lazy_adam/"):
            _close(k, sd[k[len("lazy_adam/"):]].numpy(), v)


def test_lazy_off_switch_takes_the_dense_path(mock, monkeypatch):
    monkeypatch.setenv("DCTR_LAZY_UPDATE", "0")
    g = load_golden("lazy_deepfm")
    m = build_model(g["spec"], DEV, l2=1e-3)
    m.compile("adagrad", "binary_crossentropy", metrics=[])
    assert m.model_plan().update == ("dense",)


def test_tables_are_current_when_callbacks_and_fit_return(mock, monkeypatch):
    """A callback (or user code after fit) that reads ``embedding_dict[...].weight`` directly must see the reference's
    tables: every lazily replayed row is brought up to date before on_epoch_end."""
    from deepctr_torch.callbacks import Callback
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")
    g = load_golden("lazy_deepfm")
    m = build_model(g["spec"], DEV, l2=1e-3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.compile("adam", "binary_crossentropy", metrics=[])
    X = np.concatenate(list(g["extra"]["lazy_X"]), 0)
    y = np.concatenate(list(g["extra"]["lazy_y"]), 0)
    x = {c["name"]: X[:, i] for i, c in enumerate(g["spec"]["dnn_columns"])}
    seen = []

    class Peek(Callback):
        def on_epoch_end(self, epoch, logs=None):
            seen.append(self.model.model_plan().lazy.dirty)

    m.fit(x, y, batch_size=24, epochs=2, verbose=0, shuffle=False, callbacks=[Peek()])
    assert seen == [False, False] and m.model_plan().update == ("lazy", "adam")
    # 8 unshuffled batches of 24 = the fixture's 8 steps, twice: after the first epoch the tables equal the reference's
    w = m.embedding_dict[g["spec"]["dnn_columns"][0]["embedding_name"]].weight.detach().clone()
    assert torch.equal(w, m.state_dict()["embedding_dict.%s.weight" % g["spec"]["dnn_columns"][0]["embedding_name"]])


def test_adam_tables_are_torchs_own_scalars():
    """LazyState.adam_tables: entry T - 1 = what torch.optim.Adam computes on the host at step T (adam.py: bias_correction =
    1 - beta ** step in double; step_size = lr / bias_correction1; sqrt(bias_correction2)), rounded to float32 once; the
    last entries are the limits (lr and 1) so that steps past the end may read them; a beta too close to 1 gets no table."""
    import math
    from deepctr_torch._hip.plan import LazyState
    lr, b1, b2 = 1e-3, 0.9, 0.999
    ss, bc, rbc = LazyState.adam_tables(lr, b1, b2)
    np.testing.assert_allclose(rbc.astype(np.float64) * bc.astype(np.float64), 1.0, rtol=2e-7)
    assert ss.dtype == np.float32 and bc.dtype == np.float32
    for T in (1, 2, 3, 10, 100, 349, 1000, 20000, len(bc)):
        if T <= len(ss):
            assert ss[T - 1] == np.float32(lr / (1.0 - b1 ** T)), T
        assert bc[T - 1] == np.float32(math.sqrt(1.0 - b2 ** T)), T
    assert 1.0 - b1 ** len(ss) == 1.0 and ss[-1] == np.float32(lr)
    assert 1.0 - b2 ** len(bc) == 1.0 and bc[-1] == np.float32(1.0)
    assert 340 < len(ss) < 400 and 36000 < len(bc) < 40000
    assert LazyState.adam_tables(lr, 0.0, 0.999)[0].tolist() == [np.float32(lr)]
    assert LazyState.adam_tables(lr, 0.9, 0.9999999) is None          # 370 M entries: computed in the kernel instead
    assert LazyState.adam_tables(lr, 0.9, 1.0) is None


@pytest.mark.parametrize("sweep_k", ["0", "3", "32"])
def test_sweep_changes_who_pays_not_what(mock, monkeypatch, sweep_k):
    """The per-step sweep (dctr_lazy_sweep: one K-th of every table brought to the current step in front of the catch-up)
    only moves work between the sweep, the catch-up and the flush: the same reference trajectory with K = 3 (the windows
    come by many times), the default and without."""
    monkeypatch.setenv("DCTR_LAZY_SWEEP_K", sweep_k)
    g, m, bce, tot = _run("lazy_deepfm", "adam")
    ex = g["extra"]
    assert ("lazy_sweep" in mock.calls) == (sweep_k != "0")
    np.testing.assert_allclose(bce, ex["lazy_adam_bce"], rtol=5e-5)
    sd = m.state_dict()
    for k, v in ex.items():
        if k.startswith("lazy_adam/"):
            ref = np.asarray(v)
            assert max_abs(sd[k[len("lazy_adam/"):]].numpy(), ref) <= 2e-5 * max(1.0, float(np.abs(ref).max())), k
