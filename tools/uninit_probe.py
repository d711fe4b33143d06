#!/usr/bin/env python
"""Hunt reads of uninitialised device memory: every torch.empty / empty_like / new_empty made by the package's host code
comes back filled with NaN (or a marker value); a model's outputs must not change.  When they do, the allocation sites
are bisected one by one.  Round 2: an NFM forward was 1.3e-5 off only when run after other tests (recycled blocks).
    python tools/uninit_probe.py > gpurun_out/uninit_probe.json"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "deepctr-torch_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import build_model, feature_columns, load_golden, load_matrix, matrix_id  # noqa: E402

DEV = "cuda:0"
PKG = os.path.join(ROOT, "deepctr-torch_amd")
FILL = {"value": None, "skip": set(), "seen": {}}
_empty, _empty_like = torch.empty, torch.empty_like


def _site():
    for fr in traceback.extract_stack()[::-1]:
        if fr.filename.startswith(PKG):
            return "%s:%d" % (os.path.relpath(fr.filename, PKG), fr.lineno)
    return None


def _poison(t):
    v = FILL["value"]
    if v is None or not t.is_floating_point() or not t.is_cuda:
        return t
    s = _site()
    if s is None:
        return t
    FILL["seen"][s] = FILL["seen"].get(s, 0) + 1
    if s not in FILL["skip"]:
        t.fill_(v)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))


def run_case(make, X, y, train):
    torch.manual_seed(0)
    m = make()
    out = {}
    if train:
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        for _ in range(2):
            loss, _, yp = m._train_step(X, y)
        out["loss"] = loss.detach().cpu().numpy().copy()
        out["y_pred"] = yp.detach().cpu().numpy().copy()
        for k, v in m.state_dict().items():
            out["p/" + k] = v.detach().cpu().numpy().copy()
    else:
        m.eval()
        with torch.no_grad():
            out["y_pred"] = m(X).cpu().numpy().copy()
    torch.cuda.synchronize()
    return out


def differs(a, b):
    bad = []
    for k in a:
        if not np.array_equal(a[k], b[k], equal_nan=False):
            bad.append(k)
    return bad


def probe(tag, make, X, y, train):
    FILL["value"], FILL["skip"], FILL["seen"] = None, set(), {}
    clean = run_case(make, X, y, train)
    res = {"sites": None, "culprits": [], "changed": None}
    for val in (float("nan"), 1.0e3):
        FILL["value"], FILL["skip"], FILL["seen"] = val, set(), {}
        got = run_case(make, X, y, train)
        bad = differs(clean, got)
        sites = sorted(FILL["seen"])
        res["sites"] = len(sites)
        if not bad:
            continue
        res["changed"] = bad[:6]
        # which allocation sites matter: un-poison one at a time
        for s in sites:
            FILL["value"], FILL["skip"], FILL["seen"] = val, set(x for x in sites if x != s), {}
            if differs(clean, run_case(make, X, y, train)):
                res["culprits"].append((s, "nan" if val != val else val))
        break
    FILL["value"] = None
    return res


def main():
    import deepctr_torch.models as M
    results = {}
    for c in load_matrix():
        if c["seed"] == 13:
            continue
        spec = c["spec"]
        lin, dnn = feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"])
        cls = getattr(M, c["model"])

        def make(c=c, cls=cls, lin=lin, dnn=dnn):
            m = cls(dnn, device=DEV, **c["kwargs"]) if c["model"] == "PNN" else cls(lin, dnn, device=DEV, **c["kwargs"])
            m.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()})
            return m
        X, y = torch.from_numpy(c["X"]).to(DEV), torch.from_numpy(c["y"]).to(DEV)
        for train in (False, True):
            tag = "%s/%s" % (matrix_id(c), "train" if train else "eval")
            try:
                r = probe(tag, make, X, y, train)
            except Exception as exc:  # noqa: BLE001
                r = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
            if r.get("culprits") or r.get("error") or r.get("changed"):
                results[tag] = r
    for name in ("deepfm_criteo", "xdeepfm_criteo", "fibinet_interaction", "dcn_vector", "pnn_inner", "afm_criteo"):
        try:
            g = load_golden(name)
        except Exception:  # noqa: BLE001
            continue

        def make(g=g):
            m = build_model(g["spec"], DEV)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
            return m
        X, y = torch.from_numpy(g["X"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
        for train in (False, True):
            tag = "%s/%s" % (name, "train" if train else "eval")
            try:
                r = probe(tag, make, X, y, train)
            except Exception as exc:  # noqa: BLE001
                r = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
            if r.get("culprits") or r.get("error") or r.get("changed"):
                results[tag] = r
    print(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
