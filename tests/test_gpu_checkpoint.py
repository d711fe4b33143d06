"""GPU: torch.save / load round trip of model + optimizer in the middle of training (what ModelCheckpoint-style
callbacks and resumed jobs do, reference callbacks.py:58-61,70-73): the resumed run must continue exactly where the
uninterrupted one goes -- for the fused sparse update (tables' Adagrad state, the dense slab's state) and for the
exact lazy update (row stamps are flushed into the checkpoint; Adam's step count travels in the optimizer state)."""
import io

import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(g, l2, opt):
    m = build_model(g["spec"], DEV, l2=l2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    return m


@pytest.mark.parametrize("l2,opt,mode", [(0.0, "adagrad", "adagrad"), (1e-3, "adam", "lazy"), (1e-3, "adagrad", "lazy"),
                                         (0.0, "sgd", "sgd")])
def test_resume_from_checkpoint_continues_the_trajectory(l2, opt, mode):
    g = load_golden("lazy_deepfm")
    Xs = [torch.from_numpy(x).to(DEV) for x in g["extra"]["lazy_X"]]
    ys = [torch.from_numpy(y).to(DEV) for y in g["extra"]["lazy_y"]]
    k = 4
    a = _make(g, l2, opt)
    assert a.model_plan().update[0] == mode
    for i in range(len(Xs)):
        a._train_step(Xs[i], ys[i])
    want = {n: v.clone() for n, v in a.state_dict().items()}

    b = _make(g, l2, opt)
    for i in range(k):
        b._train_step(Xs[i], ys[i])
    buf = io.BytesIO()
    torch.save({"model": b.state_dict(), "optim": b.optim.state_dict()}, buf)
    ck = torch.load(io.BytesIO(buf.getvalue()), map_location=DEV, weights_only=False)

    c = build_model(g["spec"], DEV, l2=l2)
    c.compile(opt, "binary_crossentropy", metrics=[])
    c.load_state_dict(ck["model"])
    c.optim.load_state_dict(ck["optim"])
    c.compile(c.optim, "binary_crossentropy", metrics=[])      # re-derive the update mode from the restored optimizer
    c.train()
    assert c.model_plan().update[0] == mode
    for i in range(k, len(Xs)):
        c._train_step(Xs[i], ys[i])
    got = c.state_dict()
    for n, v in want.items():
        ref = v.cpu().numpy()
        err = max_abs(got[n].cpu().numpy(), ref)
        assert err <= 1e-6 * max(1.0, float(np.abs(ref).max())), "%s: %.3e" % (n, err)
