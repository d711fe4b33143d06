"""GPU: the forward-only table layout (deepctr_torch/_hip/layout.py apply_infer_layout, round 5): a model that was never
compiled for training seats the deep row and the wide weight of an id in ONE 128-byte line.  The values the model computes do
not depend on where a row lives: predictions equal the reference's golden ones (tests/golden, produced by the real reference)
and the numpy oracle's; ``state_dict`` round-trips between the layouts; ``compile()`` afterwards re-seats for training and the
trajectory is the golden one."""
import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs
from np_oracle import Oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _named_input(m, X):
    return {name: (X[:, lo] if hi - lo == 1 else X[:, lo:hi]) for name, (lo, hi) in m.feature_index.items()}


@pytest.mark.parametrize("name", ["deepfm_criteo", "wdl_criteo", "xdeepfm_criteo"])
def test_predict_on_loaded_weights_uses_the_forward_only_layout(name):
    g = load_golden(name)
    m = build_model(g["spec"], DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    pred = m.predict(_named_input(m, g["X"]), batch_size=64)
    plan = m.model_plan()
    seated = 0
    for di, wi, _, _ in plan.units:
        if di >= 0 and wi >= 0 and plan.deep[di].dim <= 28 and plan.deep[di].dim % 4 == 0:
            pd, pw = plan.deep[di].param, plan.wide[wi].param
            assert pd.stride(0) == 32 and pw.stride(0) == 32, "deep row and wide weight share a [V, 32] slab"
            assert pw.data_ptr() == pd.data_ptr() + 4 * plan.deep[di].dim
            seated += 1
    assert seated > 0
    assert max_abs(pred.reshape(-1), g["y_pred"].reshape(-1)) <= 1e-5
    _, y64 = Oracle(g["spec"], g["params"], dtype=np.float64).forward(g["X"])
    assert max_abs(pred.reshape(-1), np.asarray(y64).reshape(-1)) <= 1e-5
    # state_dict: the reference's keys, shapes, values -- contiguous copies, nothing of the slabs
    sd = m.state_dict()
    for k, v in g["params"].items():
        assert tuple(sd[k].shape) == v.shape and sd[k].is_contiguous() and max_abs(sd[k].cpu().numpy(), v) == 0.0, k
    # ... which load into a model seated for TRAINING (interleaved with the Adagrad state) and predict the same bits
    m2 = build_model(g["spec"], DEV)
    m2.compile("adagrad", "binary_crossentropy", metrics=[])
    m2.load_state_dict(sd)
    pred2 = m2.predict(_named_input(m2, g["X"]), batch_size=64)
    assert np.array_equal(pred, pred2)
    p2 = m2.model_plan()
    assert any(p2.wide[wi].param.stride(0) == 2 for _, wi, _, _ in p2.units if wi >= 0)     # ([V, 2]: weight | Adagrad sum)


def test_compile_after_predict_reseats_for_training():
    g = load_golden("deepfm_criteo")
    m = build_model(g["spec"], DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.predict(_named_input(m, g["X"]), batch_size=64)                       # forward-only seating
    m.compile("adagrad", "binary_crossentropy", metrics=[])
    plan = m.model_plan()
    assert all(plan.wide[wi].param.stride(0) == 2 for _, wi, _, _ in plan.units if wi >= 0)
    m.train()
    losses = [m._train_step(torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV))[0].item()
              for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"])]
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses, g["extra"]["adagrad3_loss"], rtol=2e-5)
    sd = m.state_dict()
    worst = max(max_abs(sd[k[9:]].cpu().numpy(), v) for k, v in g["extra"].items() if k.startswith("adagrad3/"))
    assert worst <= 2e-5
