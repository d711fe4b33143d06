"""compile / fit / evaluate / predict call variants against the REAL reference (tests/golden/api/api_variants.npz,
oracle/make_api_golden.py): regression with mse, x as a list with validation_data, a VarLen history sharing the item table
plus a 3-wide DenseFeat, 'rmsprop', an optimizer instance with weight_decay and a loss callable.  Same History (2e-4
relative; AUC / accuracy, rank / threshold statistics over 32-48 samples: 5e-3 / one sample), same evaluate(), same
predict() and final parameters.  Runs on CPU through the numpy stand-in of the library (host logic: update-mode
selection, dict / list inputs, 2-D columns) and on the GPU through the kernels."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, feature_columns, max_abs


def _load():
    z = np.load(os.path.join(GOLDEN_DIR, "api", "api_variants.npz"), allow_pickle=False)
    out = []
    for v in json.loads(str(z["variants"])):
        pre = v["tag"] + "/"
        out.append(dict(v, data={k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}))
    return out


VARIANTS = _load()


def _model_input(m, X, as_list):
    d = {name: (X[:, lo] if hi - lo == 1 else X[:, lo:hi]) for name, (lo, hi) in m.feature_index.items()}
    return [d[k] for k in m.feature_index] if as_list else d


def _run(v, dev):
    import deepctr_torch.models as M
    d = v["data"]
    cols = feature_columns(v["cols"])
    lin = feature_columns(v["lin_cols"]) if "lin_cols" in v else cols
    m = getattr(M, v.get("model", "DeepFM"))(lin, cols, l2_reg_linear=v["l2"], l2_reg_embedding=v["l2"], device=dev,
                                               **v["kwargs"])
    m.load_state_dict({k[len("param/"):]: torch.from_numpy(val) for k, val in d.items() if k.startswith("param/")})
    if v.get("l1"):
        m.add_regularization_weight(filter(lambda kv: "weight" in kv[0], m.dnn.named_parameters()), l1=v["l1"])
        m.add_regularization_weight(m.embedding_dict["C1"].weight, l1=v["l1"])
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-4) if v["opt"] == "instance" else v["opt"]
    loss = torch.nn.functional.binary_cross_entropy if v["loss"] == "callable" else v["loss"]
    m.compile(opt, loss, metrics=v["metrics"])
    xin = _model_input(m, d["X"], v["x"] == "list")
    torch.manual_seed(5)
    if v["val"] == "data":
        hist = m.fit(xin, d["y"], batch_size=32, epochs=2, verbose=2, shuffle=False,
                     validation_data=(_model_input(m, d["Xv"], True), d["yv"]))
    else:
        hist = m.fit(xin, d["y"], batch_size=32, epochs=2, verbose=2, shuffle=False, validation_split=0.2)
    ref_hist = {k[len("hist/"):]: val for k, val in d.items() if k.startswith("hist/")}
    assert set(hist.history) == set(ref_hist)

    def close(name, got, want):
        if "auc" in name:
            np.testing.assert_allclose(got, want, atol=5e-3, err_msg=name)
        elif "acc" in name:
            np.testing.assert_allclose(got, want, atol=1.01 / 32, err_msg=name)       # one sample of a 32-row batch
        else:
            np.testing.assert_allclose(got, want, rtol=2e-4, err_msg=name)

    for k, want in ref_hist.items():
        close(k, hist.history[k], want)
    ev = m.evaluate(_model_input(m, d["Xv"], v["x"] == "list"), d["yv"], batch_size=20)
    ref_ev = {k[len("eval/"):]: val for k, val in d.items() if k.startswith("eval/")}
    assert set(ev) == set(ref_ev)
    for k, want in ref_ev.items():
        close(k, ev[k], want)
    pred = m.predict(xin, batch_size=50)
    assert pred.dtype == np.float64 and max_abs(pred, d["pred"]) <= 5e-5
    sd = m.state_dict()
    for k, val in d.items():
        if k.startswith("final/"):
            assert max_abs(sd[k[len("final/"):]].cpu().numpy(), val) <= 5e-5, k
    return m


@pytest.mark.parametrize("v", VARIANTS, ids=lambda v: v["tag"])
def test_api_variant_on_the_stand_in(mock, monkeypatch, v):
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")
    _run(v, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("v", VARIANTS, ids=lambda v: v["tag"])
def test_api_variant_on_the_gpu(v):
    _run(v, "cuda:0")


# ---- BASELINE.json configs[0]: the reference's Criteo example, end to end (oracle/check_criteo_example.py) -------------
def _criteo_example(dev):
    """examples/run_classification_criteo.py from the model definition on (the preprocessing half is checked against
    the example's pandas / sklearn pipeline by oracle/check_criteo_example.py and tests/test_data_format.py): same
    DeepFM kwargs, adagrad, 10 shuffled epochs of batch 32 with a 20 % validation split, predict on the held-out rows.
    40 Adagrad steps drive the training loss from 0.63 to 0.02; the per-epoch losses stay within 5e-4 relative of the
    reference's (measured on the numpy stand-in: 2e-7), AUC within 5e-3, the test predictions within 2e-4."""
    from sklearn.metrics import log_loss, roc_auc_score
    from deepctr_torch.inputs import DenseFeat, SparseFeat, get_feature_names
    from deepctr_torch.models import DeepFM
    z = np.load(os.path.join(GOLDEN_DIR, "api", "criteo_example.npz"), allow_pickle=False)
    names = json.loads(str(z["names"]))
    sparse, dense = names[:26], names[26:]
    cols = [SparseFeat(f, vocabulary_size=int(v), embedding_dim=4) for f, v in zip(sparse, z["vocab"])] + \
           [DenseFeat(f, 1) for f in dense]
    assert get_feature_names(cols + cols) == names
    model = DeepFM(linear_feature_columns=cols, dnn_feature_columns=cols, task='binary', l2_reg_embedding=1e-5,
                   device=dev)
    model.load_state_dict({k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")})
    model.compile("adagrad", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    train = {n: z["train_X"][:, i] for i, n in enumerate(names)}
    test = {n: z["test_X"][:, i] for i, n in enumerate(names)}
    torch.manual_seed(2020)
    hist = model.fit(train, z["train_y"], batch_size=32, epochs=10, verbose=2, validation_split=0.2)
    ref = {k[len("hist/"):]: z[k] for k in z.files if k.startswith("hist/")}
    assert set(hist.history) == set(ref)
    for k, want in ref.items():
        got = np.asarray(hist.history[k])
        if "auc" in k:
            np.testing.assert_allclose(got, want, atol=5e-3, err_msg=k)
        else:
            np.testing.assert_allclose(got, want, rtol=5e-4, err_msg=k)
    pred = model.predict(test, 256)
    assert max_abs(pred, z["pred"]) <= 2e-4
    assert abs(log_loss(z["test_y"], pred) - float(z["test_logloss"])) <= 2e-4
    assert abs(roc_auc_score(z["test_y"], pred) - float(z["test_auc"])) <= 5e-3
    return model


def test_criteo_example_on_the_stand_in(mock, monkeypatch):
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")
    m = _criteo_example("cpu")
    assert m.model_plan().update == ("lazy", "adagrad")          # the example's L2 terms, applied O(batch)


@pytest.mark.gpu
def test_criteo_example_on_the_gpu():
    _criteo_example("cuda:0")


def test_regulariser_added_after_compile_counts(mock, monkeypatch):
    """The reference evaluates ``regularization_weight`` at every step (basemodel.py:257,412-428), so a term added AFTER
    compile() takes effect; here the update path chosen at compile time (lazy / in-kernel / fused: L2-only) must be
    re-derived.  Same fixture as 'l1' -- the reference's result does not depend on the order of the two calls."""
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")
    from deepctr_torch.models import DeepFM
    v = [v for v in VARIANTS if v["tag"] == "l1"][0]
    d = v["data"]
    cols = feature_columns(v["cols"])
    m = DeepFM(cols, cols, l2_reg_linear=v["l2"], l2_reg_embedding=v["l2"], device="cpu", **v["kwargs"])
    m.load_state_dict({k[len("param/"):]: torch.from_numpy(val) for k, val in d.items() if k.startswith("param/")})
    m.compile("adagrad", "binary_crossentropy", metrics=["binary_crossentropy"])
    assert m.model_plan().update == ("lazy", "adagrad")
    m.add_regularization_weight(filter(lambda kv: "weight" in kv[0], m.dnn.named_parameters()), l1=v["l1"])
    m.add_regularization_weight(m.embedding_dict["C1"].weight, l1=v["l1"])
    assert m.model_plan().update == ("dense",)
    hist = m.fit(_model_input(m, d["X"], False), d["y"], batch_size=32, epochs=2, verbose=0, shuffle=False,
                 validation_split=0.2)
    np.testing.assert_allclose(hist.history["loss"], d["hist/loss"], rtol=2e-4)
    sd = m.state_dict()
    for k, val in d.items():
        if k.startswith("final/"):
            assert max_abs(sd[k[len("final/"):]].numpy(), val) <= 5e-5, k


# ---- learning-rate schedules stepping model.optim (tests/golden/api/lr_schedule.npz) --------------------------------------
SCHED_RUNS = (("adagrad0", "adagrad", 0.0), ("adam", "adam", 1e-5), ("sgd", "sgd", 1e-3), ("adagrad", "adagrad", 1e-3))


def _schedule(tag, opt, l2, dev):
    """9 steps of the REAL reference with lr halved after steps 3 and 6 (oracle/make_api_golden.py --schedule).  The
    O(batch) paths carry lr in kernel arguments; lazily replayed rows must be brought up to date with the OLD rate
    before the new one applies.  Total loss per step and final parameters, 5e-5 relative."""
    from deepctr_torch.models import DeepFM
    z = np.load(os.path.join(GOLDEN_DIR, "api", "lr_schedule.npz"), allow_pickle=False)
    spec = json.loads(str(z["spec"]))
    cols = feature_columns(spec["dnn_columns"])
    m = DeepFM(cols, cols, l2_reg_linear=l2, l2_reg_embedding=l2, device=dev, **spec["kwargs"])
    m.load_state_dict({k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")})
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    tot = []
    for i in range(z["X"].shape[0]):
        if i in (3, 6):
            for grp in m.optim.param_groups:
                grp["lr"] *= 0.5
        _, total, _ = m._train_step(torch.from_numpy(z["X"][i]).to(dev), torch.from_numpy(z["y"][i]).to(dev))
        tot.append(float(total))
    np.testing.assert_allclose(tot, z[tag + "/total"], rtol=5e-5)
    sd = m.state_dict()
    pre = tag + "/final/"
    for k in z.files:
        if k.startswith(pre):
            ref = z[k]
            assert max_abs(sd[k[len(pre):]].cpu().numpy(), ref) <= 5e-5 * max(1.0, float(np.abs(ref).max())), k
    return m


@pytest.mark.parametrize("tag,opt,l2", SCHED_RUNS)
def test_lr_schedule_on_the_stand_in(mock, tag, opt, l2):
    m = _schedule(tag, opt, l2, "cpu")
    assert m.model_plan().update[0] in ("lazy", "adagrad")


@pytest.mark.parametrize("tag,opt,l2", [("freeze_adagrad", "adagrad", 0.0), ("freeze_adam", "adam", 1e-5)])
def test_frozen_tables_on_the_stand_in(mock, tag, opt, l2):
    """requires_grad_(False) on one deep and one wide table (pretrained embeddings): the REAL reference leaves them
    untouched (autograd yields no gradient, torch.optim skips them) while everything else trains.  The in-kernel / lazy
    optimizers would move every table, so the exact dense-gradient route must be chosen, without a gradient for the
    frozen ones.  4 steps: total loss and every final parameter."""
    from deepctr_torch.models import DeepFM
    z = np.load(os.path.join(GOLDEN_DIR, "api", "lr_schedule.npz"), allow_pickle=False)
    spec = json.loads(str(z["spec"]))
    cols = feature_columns(spec["dnn_columns"])
    m = DeepFM(cols, cols, l2_reg_linear=l2, l2_reg_embedding=l2, device="cpu", **spec["kwargs"])
    m.load_state_dict({k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")})
    frozen = [m.embedding_dict["C1"].weight, m.linear_model.embedding_dict["C2"].weight]
    for p in frozen:
        p.requires_grad_(False)
    before = [p.detach().clone() for p in frozen]
    m.compile(opt, "binary_crossentropy", metrics=[])
    assert m.model_plan().update == ("dense",)
    m.train()
    tot = [float(m._train_step(torch.from_numpy(z["X"][i]), torch.from_numpy(z["y"][i]))[1]) for i in range(4)]
    np.testing.assert_allclose(tot, z[tag + "/total"], rtol=5e-5)
    assert all(torch.equal(p.detach(), b) for p, b in zip(frozen, before))
    sd, pre = m.state_dict(), tag + "/final/"
    for k in z.files:
        if k.startswith(pre):
            assert max_abs(sd[k[len(pre):]].numpy(), z[k]) <= 5e-5 * max(1.0, float(np.abs(z[k]).max())), k
    # thawing them brings the O(batch) path back at the next step
    for p in frozen:
        p.requires_grad_(True)
    m._train_step(torch.from_numpy(z["X"][4]), torch.from_numpy(z["y"][4]))
    assert m.model_plan().update[0] in ("adagrad", "lazy")


def test_replaced_optimizer_object_restarts_its_state(mock):
    """``model.optim = torch.optim.Adagrad(model.parameters())`` after some steps: a NEW optimizer starts from fresh
    accumulators; the in-kernel Adagrad must follow the new object's state tensors, not keep the old ones."""
    from deepctr_torch.models import DeepFM
    z = np.load(os.path.join(GOLDEN_DIR, "api", "lr_schedule.npz"), allow_pickle=False)
    spec = json.loads(str(z["spec"]))
    cols = feature_columns(spec["dnn_columns"])

    def steps(m, lo, hi):
        for i in range(lo, hi):
            m._train_step(torch.from_numpy(z["X"][i]), torch.from_numpy(z["y"][i]))

    def fresh():
        m = DeepFM(cols, cols, l2_reg_linear=0, l2_reg_embedding=0, device="cpu", **spec["kwargs"])
        m.load_state_dict({k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")})
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        return m
    a = fresh()
    steps(a, 0, 3)
    a.optim = torch.optim.Adagrad(a.parameters())           # fast paths must notice
    steps(a, 3, 6)
    import os as _os
    _os.environ["DCTR_SPARSE_UPDATE"] = "0"                  # the exact dense-gradient route as the yardstick
    try:
        b = fresh()
        steps(b, 0, 3)
        b.optim = torch.optim.Adagrad(b.parameters())
        steps(b, 3, 6)
    finally:
        del _os.environ["DCTR_SPARSE_UPDATE"]
    sa, sb = a.state_dict(), b.state_dict()
    assert a.model_plan().update[0] == "adagrad" and b.model_plan().update == ("dense",)
    for k in sa:
        assert max_abs(sa[k].numpy(), sb[k].numpy()) <= 2e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("tag,opt,l2", SCHED_RUNS)
def test_lr_schedule_on_the_gpu(tag, opt, l2):
    _schedule(tag, opt, l2, "cuda:0")


def test_same_seed_same_initial_weights_as_the_reference():
    """Model construction consumes the torch generator exactly like the reference (same modules, same order, same
    init functions): DeepFM(..., seed=1024) of the Criteo example starts from the reference's own initial state_dict,
    bit for bit -- so a user's run is reproducible across the two packages without copying weights."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    z = np.load(os.path.join(GOLDEN_DIR, "api", "criteo_example.npz"), allow_pickle=False)
    names = json.loads(str(z["names"]))
    cols = [SparseFeat(f, vocabulary_size=int(v), embedding_dim=4) for f, v in zip(names[:26], z["vocab"])] + \
           [DenseFeat(f, 1) for f in names[26:]]
    m = DeepFM(linear_feature_columns=cols, dnn_feature_columns=cols, task='binary', l2_reg_embedding=1e-5, device="cpu")
    sd = m.state_dict()
    ref = {k[len("param/"):]: z[k] for k in z.files if k.startswith("param/")}
    assert set(sd) == set(ref)
    for k, v in ref.items():
        assert np.array_equal(sd[k].numpy(), v), k


# ---- every model family FROM SCRATCH (tests/golden/api/scratch_runs.json = the reference's side of oracle/diff_api.py) ------
def _scratch_setup():
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    rng = np.random.default_rng(0)
    N = 100
    x = {"a": rng.integers(0, 10, N), "b": rng.integers(0, 7, N), "d": rng.random(N)}
    y = rng.integers(0, 2, N)
    cols = [SparseFeat("a", 10, 4), SparseFeat("b", 7, 4), SparseFeat("c", 5, 4), DenseFeat("d", 1), DenseFeat("e", 1),
            VarLenSparseFeat(SparseFeat("h", 9, 4), 3, "mean")]
    x = dict(x, c=rng.integers(0, 5, N), e=rng.random(N), h=rng.integers(0, 9, (N, 3)))
    return cols, x, y


SCRATCH = {
    "DeepFM": dict(dnn_hidden_units=(8, 4)), "xDeepFM": dict(dnn_hidden_units=(8,), cin_layer_size=(6, 4)),
    "FiBiNET": dict(dnn_hidden_units=(8,)), "DCN": dict(dnn_hidden_units=(8,), cross_num=2),
    "DCN_matrix": dict(dnn_hidden_units=(8,), cross_num=2, cross_parameterization="matrix"),
    "DCNMix": dict(dnn_hidden_units=(8,), cross_num=2, low_rank=4, num_experts=2), "PNN": dict(dnn_hidden_units=(8,)),
    "PNN_outer": dict(dnn_hidden_units=(8,), use_outter=True), "NFM": dict(dnn_hidden_units=(8,)),
    "AFM": dict(attention_factor=4), "AutoInt": dict(dnn_hidden_units=(8,), att_layer_num=2), "WDL": dict(dnn_hidden_units=(8,)),
}


def _scratch(name, dev):
    """Nothing is copied from the reference here: the model is CONSTRUCTED with the default seed, compiled with adam and
    the default L2, fitted for two shuffled epochs after torch.manual_seed(3) and asked to predict -- and must print the
    reference's History (2e-4) and predictions, because construction, shuffling and every step consume and compute the
    same things."""
    import deepctr_torch.models as MM
    from deepctr_torch.inputs import DenseFeat
    ref = json.load(open(os.path.join(GOLDEN_DIR, "api", "scratch_runs.json")))["scratch_" + name]
    cols, x, y = _scratch_setup()
    cols_ = [c for c in cols if not isinstance(c, DenseFeat)] if name == "AFM" else cols
    cls = getattr(MM, name.split("_")[0])
    kw = SCRATCH[name]
    m = cls(cols_, device=dev, **kw) if name.startswith("PNN") else cls(cols_, cols_, device=dev, **kw)
    m.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy"])
    torch.manual_seed(3)
    xin = {k: v for k, v in x.items() if k in m.feature_index}
    h = m.fit(xin, y, batch_size=32, epochs=2, verbose=0, validation_split=0.2)
    assert set(h.history) == set(ref["hist"])
    for k, v in ref["hist"].items():
        np.testing.assert_allclose(h.history[k], v, rtol=2e-4, atol=2e-5, err_msg=k)
    assert abs(float(m.predict(xin, 64).sum()) - ref["pred_sum"]) <= 2e-3


@pytest.mark.parametrize("name", sorted(SCRATCH))
def test_from_scratch_run_matches_the_reference_on_the_stand_in(mock, monkeypatch, name):
    monkeypatch.setenv("DCTR_FIT_GRAPH", "0")
    _scratch(name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCRATCH))
def test_from_scratch_run_matches_the_reference_on_the_gpu(name):
    _scratch(name, "cuda:0")
