"""GPU: the reference's OWN model tests, restated (tests/models/{DeepFM,xDeepFM,FiBiNET,DCN,DCNMix,PNN,NFM,AFM,AutoInt,
WDL}_test.py with tests/utils.py:19-66 ``get_test_data`` and :142-171 ``check_model``) -- same parameter matrices, same
kind of data (1-9 row vocabularies, so vocabulary 1 and id 0 occur; sum / mean / max VarLen columns of random maxlen
with 0 as padding id; 64 samples), same protocol: compile('adam', 'binary_crossentropy', ['binary_crossentropy',
'acc']), one epoch of fit with batch 100 > sample count, validation_split 0.5, EarlyStopping + ModelCheckpoint on
'val_acc', state_dict save / load, whole-model torch.save / torch.load.

The reference's tests assert nothing numeric; here every case ALSO checks the eval-mode forward against the
numpy oracle on the model's own initial parameters (2e-5), wherever the oracle states the configuration (not PReLU
towers), and that History / predict have the reference's shape."""
import os

import numpy as np
import pytest
import torch

from helpers import max_abs

from matrix_data import N, clean_rows, make_data, spec_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def check_model(model, x, y, tmp_path, spec=None, cols=()):
    from deepctr_torch.callbacks import EarlyStopping, ModelCheckpoint
    X = model._as_matrix([x[name] for name in model.feature_index])
    ok = clean_rows(x, cols)
    if spec is not None and ok.any():
        from np_oracle import Oracle
        params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            got = model(X).cpu().numpy()
        _, want = Oracle(spec, params, dtype=np.float64).forward(X.cpu().numpy())
        assert max_abs(got[ok], np.asarray(want)[ok]) <= 2e-5
    ckpt = str(tmp_path / "model.ckpt")
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "acc"])
    hist = model.fit(x, y, batch_size=100, epochs=1, validation_split=0.5, verbose=2, callbacks=[
        EarlyStopping(monitor="val_acc", min_delta=0, verbose=1, patience=0, mode="max"),
        ModelCheckpoint(filepath=ckpt, monitor="val_acc", verbose=1, save_best_only=True, save_weights_only=False,
                        mode="max", period=1)])
    assert set(hist.history) == {"loss", "binary_crossentropy", "acc", "val_binary_crossentropy", "val_acc"}
    assert all(np.isfinite(v).all() for v in hist.history.values())
    assert os.path.exists(ckpt)
    w = str(tmp_path / "weights.h5")
    torch.save(model.state_dict(), w)
    model.load_state_dict(torch.load(w))
    before = model.predict(x, batch_size=50)
    assert before.shape == (N, 1) and before.dtype == np.float64 and np.isfinite(before).all()
    f = str(tmp_path / "model.h5")
    torch.save(model, f)
    again = torch.load(f, weights_only=False)
    assert max_abs(again.predict(x, batch_size=50), before) == 0.0
    return model


@pytest.mark.parametrize("use_fm,hidden,n_sparse,n_dense", [
    (True, (32,), 3, 3), (False, (32,), 3, 3), (False, (32,), 2, 2), (False, (32,), 1, 1), (True, (), 1, 1),
    (False, (), 2, 2), (True, (32,), 0, 3), (True, (32,), 3, 0), (False, (32,), 0, 3), (False, (32,), 3, 0)])
def test_DeepFM(tmp_path, use_fm, hidden, n_sparse, n_dense):
    from deepctr_torch.models import DeepFM
    x, y, cols = make_data(1, n_sparse, n_dense)
    kw = dict(use_fm=use_fm, dnn_hidden_units=hidden)
    check_model(DeepFM(cols, cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path, spec_of("DeepFM", cols, cols, **kw), cols=cols)
    check_model(DeepFM([], cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path, spec_of("DeepFM", [], cols, **kw), cols=cols)


def test_rows_of_nothing_but_padding_under_max_pooling_run(tmp_path):
    """Vocabulary 1 under 'max': EVERY row pools to embedding - 1e9 (see ``clean_rows``).  No numeric claim -- the
    reference's own value is rounding noise -- but fit / predict / save / load must go through, as in the reference."""
    from deepctr_torch.models import DeepFM
    x, y, cols = make_data(1, 0, 3, min_clean=0)
    assert not clean_rows(x, cols).any()
    check_model(DeepFM(cols, cols, dnn_hidden_units=(32,), dnn_dropout=0.5, device=DEV), x, y, tmp_path, cols=cols)


@pytest.mark.parametrize("seqs", [("sum", "mean"), ("sum", "mean", "max")])
def test_DeepFM_with_sequence_lengths(tmp_path, seqs):
    """``length_name`` columns (tests/utils.py:58-60).  With 'max' the reference itself raises on current torch
    (sequence.py:66 subtracts a bool mask); the drop-in computes the masked max the layer intends (oracle)."""
    from deepctr_torch.models import DeepFM
    x, y, cols = make_data(2, 2, 2, include_length=True, seqs=seqs)
    kw = dict(dnn_hidden_units=(32,))
    check_model(DeepFM(cols, cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path, spec_of("DeepFM", cols, cols, **kw), cols=cols)


@pytest.mark.parametrize("hidden,cin,split_half,act,n_sparse", [
    ((), (), True, "linear", 1), ((8,), (), True, "linear", 1), ((), (8,), True, "linear", 2),
    ((8,), (8,), False, "relu", 2)])
def test_xDeepFM(tmp_path, hidden, cin, split_half, act, n_sparse):
    from deepctr_torch.models import xDeepFM
    x, y, cols = make_data(3, n_sparse, n_sparse)
    kw = dict(dnn_hidden_units=hidden, cin_layer_size=cin, cin_split_half=split_half, cin_activation=act)
    check_model(xDeepFM(cols, cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path,
                spec_of("xDeepFM", cols, cols, **kw), cols=cols)


@pytest.mark.parametrize("bilinear_type", ["each", "interaction", "all"])
def test_FiBiNET(tmp_path, bilinear_type):
    from deepctr_torch.models import FiBiNET
    x, y, cols = make_data(4, 3, 3)
    kw = dict(bilinear_type=bilinear_type, dnn_hidden_units=[8, 8])
    check_model(FiBiNET(cols, cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path,
                spec_of("FiBiNET", cols, cols, **kw), cols=cols)


@pytest.mark.parametrize("cross_num,hidden,n_sparse,param", [(2, (32,), 2, "vector"), (1, (32,), 2, "matrix")])
def test_DCN(tmp_path, cross_num, hidden, n_sparse, param):
    from deepctr_torch.models import DCN
    x, y, cols = make_data(5, n_sparse, n_sparse)
    kw = dict(cross_num=cross_num, cross_parameterization=param, dnn_hidden_units=hidden)
    check_model(DCN(linear_feature_columns=cols, dnn_feature_columns=cols, dnn_dropout=0.5, device=DEV, **kw), x, y,
                tmp_path, spec_of("DCN", cols, cols, **kw), cols=cols)


@pytest.mark.parametrize("cross_num,hidden,n_sparse", [(0, (32,), 2), (1, (32,), 2)])
def test_DCNMix(tmp_path, cross_num, hidden, n_sparse):
    from deepctr_torch.models import DCNMix
    x, y, cols = make_data(6, n_sparse, n_sparse)
    kw = dict(cross_num=cross_num, dnn_hidden_units=hidden)
    check_model(DCNMix(linear_feature_columns=cols, dnn_feature_columns=cols, dnn_dropout=0.5, device=DEV, **kw), x, y,
                tmp_path, spec_of("DCNMix", cols, cols, **kw) if cross_num else None, cols=cols)


@pytest.mark.parametrize("use_inner,use_outter,kernel_type,n_sparse", [
    (True, True, "mat", 2), (True, False, "mat", 2), (False, True, "vec", 3), (False, True, "num", 3),
    (False, False, "mat", 1)])
def test_PNN(tmp_path, use_inner, use_outter, kernel_type, n_sparse):
    from deepctr_torch.models import PNN
    x, y, cols = make_data(7, n_sparse, n_sparse)
    kw = dict(dnn_hidden_units=[32, 32], use_inner=use_inner, use_outter=use_outter, kernel_type=kernel_type)
    check_model(PNN(cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path, spec_of("PNN", [], cols, **kw), cols=cols)


@pytest.mark.parametrize("n_sparse", [2, 1])
def test_NFM(tmp_path, n_sparse):
    from deepctr_torch.models import NFM
    x, y, cols = make_data(8, n_sparse, n_sparse)
    kw = dict(dnn_hidden_units=[32, 32])
    check_model(NFM(cols, cols, dnn_dropout=0.5, device=DEV, **kw), x, y, tmp_path, spec_of("NFM", cols, cols, **kw), cols=cols)


def test_AFM(tmp_path):
    from deepctr_torch.callbacks import EarlyStopping, ModelCheckpoint
    from deepctr_torch.models import AFM
    x, y, cols = make_data(9, 3, 0)
    kw = dict(use_attention=True)
    model = check_model(AFM(linear_feature_columns=cols, dnn_feature_columns=cols, afm_dropout=0.5, device=DEV, **kw),
                        x, y, tmp_path, spec_of("AFM", cols, cols, **kw), cols=cols)
    # the reference's AFM test goes on: two more fits of 3 epochs monitored on val_binary_crossentropy (AFM_test.py:25-38)
    stop = EarlyStopping(monitor="val_binary_crossentropy", min_delta=0, verbose=1, patience=0, mode="min")
    for best_only in (True, False):
        ck = ModelCheckpoint(filepath=str(tmp_path / "m.ckpt"), monitor="val_binary_crossentropy", verbose=1,
                             save_best_only=best_only, save_weights_only=False, mode="max", period=1)
        hist = model.fit(x, y, batch_size=64, epochs=3, validation_split=0.5, verbose=2, callbacks=[stop, ck])
        assert 1 <= len(hist.history["loss"]) <= 3


@pytest.mark.parametrize("att_layer_num,hidden,n_sparse", [(1, (4,), 2), (0, (4,), 2), (2, (4, 4), 2), (1, (), 1),
                                                           (1, (4,), 1)])
def test_AutoInt(tmp_path, att_layer_num, hidden, n_sparse):
    from deepctr_torch.models import AutoInt
    x, y, cols = make_data(10, n_sparse, n_sparse)
    kw = dict(att_layer_num=att_layer_num, dnn_hidden_units=hidden)
    check_model(AutoInt(linear_feature_columns=cols, dnn_feature_columns=cols, dnn_dropout=0.5, device=DEV, **kw), x, y,
                tmp_path, spec_of("AutoInt", cols, cols, **kw) if att_layer_num else None, cols=cols)


@pytest.mark.parametrize("n_sparse,n_dense", [(2, 0), (0, 2), (2, 2)])
def test_WDL(tmp_path, n_sparse, n_dense):
    from deepctr_torch.models import WDL
    x, y, cols = make_data(11, n_sparse, n_dense)
    check_model(WDL(cols, cols, dnn_activation="prelu", dnn_hidden_units=[32, 32], dnn_dropout=0.5, device=DEV), x, y,
                tmp_path, cols=cols)          # (PReLU tower: not stated by the numpy oracle -- protocol checks only)


# ---- the HIP forward against the REFERENCE's stored predictions on the same matrix (oracle/check_matrix.py) ----------
from helpers import feature_columns, load_matrix, matrix_id  # noqa: E402

GPU_MATRIX = load_matrix()


@pytest.mark.parametrize("c", GPU_MATRIX, ids=matrix_id)
def test_forward_matches_reference_on_its_test_matrix(c):
    """The reference's own freshly initialised parameters (state_dict loaded as is) and inputs for every configuration
    of its model tests: eval-mode logits of the HIP path within 1e-5 of the reference's on every row with a defined
    value; the all-padding 'max' rows must still come out finite."""
    import deepctr_torch.models as M
    spec = c["spec"]
    lin, dnn = feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"])
    cls = getattr(M, c["model"])
    m = cls(dnn, device=DEV, **c["kwargs"]) if c["model"] == "PNN" else cls(lin, dnn, device=DEV, **c["kwargs"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()})
    m.eval()
    cap = {}
    hook = m.out.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("logit", inp[0].detach()))
    with torch.no_grad():
        got = m(torch.from_numpy(c["X"]).to(DEV)).cpu().numpy()
    hook.remove()
    assert np.isfinite(got).all()
    ok = c["clean"]
    assert max_abs(cap["logit"].cpu().numpy().reshape(-1, 1)[ok], c["logit"][ok]) <= 1e-5      # north_star's bound
    assert max_abs(got[ok], c["y_pred"][ok]) <= 5e-6


@pytest.mark.parametrize("c", GPU_MATRIX, ids=matrix_id)
def test_gradients_match_reference_on_its_test_matrix(c):
    """d BCE(sum over the rows with a defined value) / d every parameter == the reference's autograd, 2e-5 x max|g|:
    the backward of sum / mean / max pooling with padding, one-row vocabularies, towers of zero layers, ...
    The two train-mode BatchNorm towers are ill-conditioned at initialisation (64 rows of ~1e-4-sized activations divided
    by sqrt(var + 1e-5)): the reference's OWN fp32 gradients are 3-10 % away from the same reference evaluated with an
    fp64 tower (stored as grad64 by oracle/check_matrix.py).  There the bound per parameter is 4 x that measured fp32
    uncertainty, against the fp64 value -- the tightest statement the reference itself supports."""
    import deepctr_torch.models as M
    spec = c["spec"]
    lin, dnn = feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"])
    cls = getattr(M, c["model"])
    m = cls(dnn, device=DEV, **c["kwargs"]) if c["model"] == "PNN" else cls(lin, dnn, device=DEV, **c["kwargs"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()})
    m.train()
    ok = torch.from_numpy(c["clean"]).to(DEV)
    y = torch.from_numpy(c["y"]).to(DEV)
    m.zero_grad()
    torch.nn.functional.binary_cross_entropy(m(torch.from_numpy(c["X"]).to(DEV)).squeeze(1)[ok], y[ok],
                                             reduction="sum").backward()
    m.model_plan().check_ids()
    for k, p in m.named_parameters():
        ref = c["grads"][k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        tol = 2e-5 * max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
        if k in c["grads64"]:
            ref64 = c["grads64"][k]
            tol = max(tol, 4.0 * max_abs(ref, ref64))
            ref = ref64
        assert max_abs(got, ref) <= tol, "%s: %.3e > %.3e (grad is None: %s, requires_grad: %s, sparse update: %s)" % (
            k, max_abs(got, ref), tol, p.grad is None, p.requires_grad, m.model_plan().update[0])
