"""Differential probe of the user-facing API: the same script runs under the REAL reference (torch-CPU) and under the
drop-in (its Python stack on CPU tensors over tests/mock_lib.py) and everything observable is compared -- History values,
the text fit() prints, exception types and messages, metric names, defaults, predict() dtype / shape.  Build container
only (needs /root/reference); nothing here is imported by the product or by the GPU tests.

    python oracle/diff_api.py ref  > /tmp/ref.json ; python oracle/diff_api.py mine > /tmp/mine.json ; diff them
    python oracle/diff_api.py            # runs both in subprocesses and reports
"""
import sys, json, io, contextlib, os
which = sys.argv[1] if len(sys.argv) > 1 else "both"
import numpy as np
if which == "both":
    import subprocess
    res = {}
    for w in ("ref", "mine"):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), w], capture_output=True, text=True, cwd="/tmp").stdout
        res[w] = json.loads([l for l in o.splitlines() if l.startswith("JSON")][-1][4:])
    bad = 0
    for k in res["ref"]:
        same = res["ref"][k] == res["mine"].get(k)
        bad += not same
        print("%-16s %s" % (k, "same" if same else "DIFFERENT\n  ref : %s\n  mine: %s" % (res["ref"][k], res["mine"].get(k))))
    sys.exit(1 if bad else 0)
import torch
if which == "ref":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden as mg
    mg.import_reference()
else:
    _root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(_root, "tests"), os.path.join(_root, "deepctr-torch_amd"), os.path.join(_root, "oracle")]
    os.environ["DCTR_FIT_GRAPH"] = "0"
    from _pytest.monkeypatch import MonkeyPatch
    mp = MonkeyPatch()
    from deepctr_torch._hip import lib as L
    from mock_lib import MockLib
    mk = MockLib()
    mp.setattr(L, "lib", lambda: mk); mp.setattr(L, "require_gpu", lambda t, what: None); mp.setattr(L, "stream_handle", lambda device=None: None)
    mp.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
from deepctr_torch.inputs import SparseFeat, DenseFeat
from deepctr_torch.models import DeepFM
rng = np.random.default_rng(0)
cols = [SparseFeat("a", 10, 4), SparseFeat("b", 7, 4), DenseFeat("d", 1)]
N = 100
x = {"a": rng.integers(0, 10, N), "b": rng.integers(0, 7, N), "d": rng.random(N)}
y = rng.integers(0, 2, N)
out = {}
def run(name, fn):
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            r = fn()
        out[name] = {"stdout": [l for l in buf.getvalue().splitlines() if "pypi" not in l], "result": r}
    except Exception as e:
        out[name] = {"error": type(e).__name__, "msg": str(e)[:80]}
def mk_model(**kw):
    torch.manual_seed(0)
    m = DeepFM(cols, cols, dnn_hidden_units=(8,), device="cpu", **kw)
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for k, p in sorted(m.state_dict().items()):
            if p.dtype.is_floating_point: p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return m
def a():
    m = mk_model(); m.compile("adagrad", "binary_crossentropy", metrics=["logloss", "auc", "acc", "mse"])
    h = m.fit(x, y, batch_size=32, epochs=2, verbose=2, validation_split=0.2, shuffle=False)
    return {k: [round(float(v), 5) for v in vs] for k, vs in h.history.items()}
def b():
    m = mk_model(); m.compile("sgd", "binary_crossentropy")
    h = m.fit(x, y, epochs=1, verbose=1, shuffle=False)
    return {k: [round(float(v), 5) for v in vs] for k, vs in h.history.items()}
def c():
    m = mk_model(); m.compile("nadam", "binary_crossentropy")
def d():
    m = mk_model(); m.compile("sgd", "hinge")
def e():
    m = mk_model(); m.compile("sgd", "binary_crossentropy", metrics=["f1"])
    return sorted(m.metrics)
def f():
    m = mk_model(); m.compile("sgd", "binary_crossentropy", metrics=["acc"])
    return m.fit(x, y, batch_size=64, epochs=1, verbose=0, validation_data=(x, y, None), shuffle=False).history.keys().__len__()
def gq():
    m = mk_model(); m.compile("sgd", "binary_crossentropy")
    return m.fit(x, y, batch_size=64, epochs=1, verbose=0, validation_data=(x,), shuffle=False)
def h_():
    m = mk_model(task="regression"); m.compile("adam", "mae", metrics=["mse"])
    hh = m.fit(x, y.astype(float), batch_size=50, epochs=1, verbose=2, shuffle=False)
    p = m.predict(x, 30)
    return [round(float(v), 5) for v in hh.history["loss"]] + [list(p.shape), str(p.dtype), round(float(p.sum()), 4)]
def i_():
    m = mk_model(task="multiclass")
# ---- every model family FROM SCRATCH: seeded construction, adam with the default L2, two shuffled epochs -------------------
from deepctr_torch.inputs import VarLenSparseFeat
import deepctr_torch.models as MM
fam_cols = [SparseFeat("a", 10, 4), SparseFeat("b", 7, 4), SparseFeat("c", 5, 4), DenseFeat("d", 1), DenseFeat("e", 1),
            VarLenSparseFeat(SparseFeat("h", 9, 4), 3, "mean")]
fam_x = dict(x, c=rng.integers(0, 5, N), e=rng.random(N), h=rng.integers(0, 9, (N, 3)))
FAMILIES = {
    "DeepFM": dict(dnn_hidden_units=(8, 4)), "xDeepFM": dict(dnn_hidden_units=(8,), cin_layer_size=(6, 4)),
    "FiBiNET": dict(dnn_hidden_units=(8,)), "DCN": dict(dnn_hidden_units=(8,), cross_num=2),
    "DCN_matrix": dict(dnn_hidden_units=(8,), cross_num=2, cross_parameterization="matrix"),
    "DCNMix": dict(dnn_hidden_units=(8,), cross_num=2, low_rank=4, num_experts=2), "PNN": dict(dnn_hidden_units=(8,)),
    "PNN_outer": dict(dnn_hidden_units=(8,), use_outter=True), "NFM": dict(dnn_hidden_units=(8,)),
    "AFM": dict(attention_factor=4), "AutoInt": dict(dnn_hidden_units=(8,), att_layer_num=2), "WDL": dict(dnn_hidden_units=(8,)),
    # outside the tower kernels: a 1024-wide layer, BatchNorm, PReLU
    "DeepFM_wide": dict(dnn_hidden_units=(1024, 8)), "DeepFM_bn": dict(dnn_hidden_units=(8,), dnn_use_bn=True),
    "WDL_prelu": dict(dnn_hidden_units=(8,), dnn_activation="prelu"),
}
def family(name, kw):
    def fn():
        cls = getattr(MM, name.split("_")[0])
        cols_ = [c for c in fam_cols if not isinstance(c, DenseFeat)] if name == "AFM" else fam_cols
        m = cls(cols_, device="cpu", **kw) if name.startswith("PNN") else cls(cols_, cols_, device="cpu", **kw)
        m.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy"])
        torch.manual_seed(3)
        h = m.fit({k: v for k, v in fam_x.items() if k in m.feature_index}, y, batch_size=32, epochs=2, verbose=0,
                  validation_split=0.2)
        p = m.predict({k: v for k, v in fam_x.items() if k in m.feature_index}, 64)
        return {"hist": {k: [round(float(v), 5) for v in vs] for k, vs in h.history.items()}, "pred_sum": round(float(p.sum()), 4)}
    return fn
for n, kw in FAMILIES.items():
    run("scratch_" + n, family(n, kw))
def pandas_inputs():
    import pandas as pd
    df = pd.DataFrame({"a": x["a"], "b": x["b"], "d": x["d"], "label": y})
    m = mk_model(); m.compile("adagrad", "binary_crossentropy", metrics=["auc"])
    xin = {n_: df[n_] for n_ in ("a", "b", "d")}                       # pandas Series, as in the examples
    h = m.fit(xin, df[["label"]].values, batch_size=32, epochs=2, verbose=0, validation_split=0.2, shuffle=False)
    p1 = m.predict(xin, 40)
    first = {k: [round(float(v), 5) for v in vs] for k, vs in h.history.items()}
    h2 = m.fit([df["a"].values, df["b"].values, df["d"].values], df["label"].values, batch_size=50, epochs=1, verbose=0, shuffle=False)
    ev = m.evaluate(xin, df["label"].values, 30)
    return [first, round(float(p1.sum()), 4),
            [round(float(v), 5) for v in h2.history["loss"]], {k: round(float(v), 5) for k, v in ev.items()}]
run("pandas_inputs", pandas_inputs)
for n, fn in (("metrics", a), ("defaults", b), ("bad_opt", c), ("bad_loss", d), ("unknown_metric", e), ("val3", f), ("val1", gq), ("mae", h_), ("bad_task", i_)):
    run(n, fn)
print("JSON" + json.dumps(out, sort_keys=True, default=str))
