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
    from deepctr_torch.models import DeepFM
    d = v["data"]
    cols = feature_columns(v["cols"])
    m = DeepFM(cols, cols, l2_reg_linear=v["l2"], l2_reg_embedding=v["l2"], device=dev, **v["kwargs"])
    m.load_state_dict({k[len("param/"):]: torch.from_numpy(val) for k, val in d.items() if k.startswith("param/")})
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
