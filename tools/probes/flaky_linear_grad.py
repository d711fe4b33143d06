#!/usr/bin/env python
"""Probe for an order-dependent failure of tests/test_gpu_reference_matrix.py::test_gradients_match_reference_on_its_test_matrix
(linear_model.weight gradient came back as zeros on xDeepFM-2s2d): runs the matrix's gradient check in the test's order,
prints what the failing case looks like and whether an immediate re-run of the same case fails again."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import feature_columns, load_matrix, matrix_id  # noqa: E402

DEV = "cuda:0"
import deepctr_torch.models as M  # noqa: E402


def run(c, verbose=False):
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
    bad = []
    for k, p in m.named_parameters():
        ref = c["grads"][k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        tol = 2e-5 * max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
        if k in c["grads64"]:
            tol = max(tol, 4.0 * float(np.max(np.abs(ref - c["grads64"][k]))) if ref.size else tol)
            ref = c["grads64"][k]
        err = float(np.max(np.abs(got - ref))) if ref.size else 0.0
        if err > tol:
            plan = m.model_plan()
            bad.append((k, err, tol, p.grad is None, p.requires_grad,
                        plan.wide_dense_weight is getattr(m.linear_model, "weight", None), plan.update[0],
                        got.reshape(-1)[:4].tolist(), ref.reshape(-1)[:4].tolist()))
    if bad and verbose is not None:
        Xd = torch.from_numpy(c["X"]).to(DEV)
        with torch.enable_grad():
            parts = m.logit_parts(Xd)
        print("   train-mode parts:", [(tuple(q.shape), float(q.abs().max()), bool(torch.isfinite(q).all())) for q in parts])
        m.eval()
        with torch.no_grad():
            parts = m.logit_parts(Xd)
        print("   eval-mode parts: ", [(tuple(q.shape), float(q.abs().max()), bool(torch.isfinite(q).all())) for q in parts])
        print("   ref logit max:", float(np.abs(c["logit"]).max()))
    return bad


cases = [c for c in load_matrix() if any(s in matrix_id(c) for s in ("FiBiNET", "xDeepFM", "DCN", "PNN"))]
n_bad = 0
for rep in range(3):
    for c in cases:
        bad = run(c)
        if bad:
            n_bad += 1
            print("rep", rep, matrix_id(c), bad[:2])
            again = run(c)
            print("   immediate re-run:", "fails again" if again else "passes", again[:1])
print("cases", len(cases), "failures", n_bad)
