"""GPU: model.fit() against the REAL reference's fit() (tests/golden/fit_deepfm.npz, oracle/make_golden.py): 3 epochs
over 225 train rows in batches of 64 (a ragged last batch), 25 % validation split, per-batch metrics averaged over steps
(basemodel.py:264-280), evaluate() at every epoch end, predict() afterwards.  Three runs: adagrad without L2 unshuffled;
the same shuffled after torch.manual_seed (the drop-in draws the DataLoader's permutations); the reference's DEFAULT
kwargs (l2 = 1e-5, adam) shuffled -- which takes the exact lazy update (csrc/lazy.hip) under hipGraph replays.
Models: DeepFM (fused train step), DCN (autograd + torch.optim around the kernels), xDeepFM (CIN on MFMA).
Tolerance: 2e-4 relative on the losses, 5e-3 absolute on AUC (a rank statistic over 33-75 samples: one swapped pair of
near-equal predictions moves it by ~1e-3)."""
import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIT_RUNS = (("plain", "adagrad", 0.0, False), ("shuffled", "adagrad", 0.0, True), ("default", "adam", 1e-5, True))


@pytest.mark.parametrize("graphs", ["1", "0"])
@pytest.mark.parametrize("tag,opt,l2,shuffle", FIT_RUNS)
@pytest.mark.parametrize("name", ["fit_deepfm", "fit_dcn", "fit_xdeepfm"])
def test_fit_history_and_predict_match_reference(monkeypatch, name, tag, opt, l2, shuffle, graphs):
    monkeypatch.setenv("DCTR_FIT_GRAPH", graphs)
    g = load_golden(name)
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
        if k.endswith("auc"):
            np.testing.assert_allclose(hist.history[k], v, atol=5e-3, err_msg=k)
        else:
            np.testing.assert_allclose(hist.history[k], v, rtol=2e-4, err_msg=k)
    pred = m.predict(x, batch_size=50)
    assert pred.dtype == np.float64 and pred.shape == ex["fit_%s_pred" % tag].shape
    assert max_abs(pred, ex["fit_%s_pred" % tag]) <= 5e-5
