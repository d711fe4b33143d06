"""Differential probe of the reference-shaped helpers of inputs.py / BaseModel (create_embedding_matrix, embedding_lookup,
varlen_embedding_lookup, get_varlen_pooling_list, get_dense_input, maxlen_lookup, combined_dnn_input,
input_from_feature_columns, linear_model) on a schema with a shared table, a length column and a 2-wide DenseFeat: the REAL
reference vs the drop-in on the stand-in library.  Build container only.

    python oracle/diff_inputs.py          # runs both in subprocesses and reports
"""
import sys, json, os
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which == "both":
    import subprocess
    res = {}
    for w in ("ref", "mine"):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), w], capture_output=True, text=True, cwd="/tmp").stdout
        res[w] = json.loads([l for l in o.splitlines() if l.startswith("JSON")][-1][4:])
    bad = [k for k in res["ref"] if res["ref"][k] != res["mine"].get(k)]
    for k in res["ref"]:
        print("%-36s %s" % (k, "DIFFERENT  ref %s  mine %s" % (res["ref"][k], res["mine"].get(k)) if k in bad else "same"))
    sys.exit(1 if bad else 0)
import numpy as np, torch
if which == "ref":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden as mg
    mg.import_reference()
else:
    _root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(_root, "tests"), os.path.join(_root, "deepctr-torch_amd"), os.path.join(_root, "oracle")]
    from _pytest.monkeypatch import MonkeyPatch
    mp = MonkeyPatch()
    from deepctr_torch._hip import lib as L
    from mock_lib import MockLib
    mk = MockLib()
    mp.setattr(L, "lib", lambda: mk); mp.setattr(L, "require_gpu", lambda t, what: None); mp.setattr(L, "stream_handle", lambda device=None: None)
    mp.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
import deepctr_torch.inputs as I
from deepctr_torch.inputs import SparseFeat, DenseFeat, VarLenSparseFeat
from deepctr_torch.models import DeepFM
cols = [SparseFeat("u", 9, 4), SparseFeat("i", 11, 4), DenseFeat("d", 2),
        VarLenSparseFeat(SparseFeat("h", 11, 4, embedding_name="i"), 3, "mean", "hl"),
        VarLenSparseFeat(SparseFeat("g", 6, 4), 2, "sum")]
fi = I.build_input_features(cols)
rng = np.random.default_rng(5)
B = 7
X = np.zeros((B, max(v[1] for v in fi.values())), np.float32)
X[:, fi["u"][0]] = rng.integers(0, 9, B); X[:, fi["i"][0]] = rng.integers(0, 11, B)
X[:, fi["d"][0]:fi["d"][1]] = rng.random((B, 2))
X[:, fi["h"][0]:fi["h"][1]] = rng.integers(1, 11, (B, 3)); X[:, fi["hl"][0]] = rng.integers(1, 4, B)
X[:, fi["g"][0]:fi["g"][1]] = rng.integers(0, 6, (B, 2))
Xt = torch.from_numpy(X)
def seeded(mod):
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for k, p in sorted(mod.state_dict().items()):
            if p.dtype.is_floating_point: p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    return mod
def desc(t):
    if torch.is_tensor(t): return {"shape": list(t.shape), "sum": round(float(t.double().sum()), 4)}
    if isinstance(t, (list, tuple)): return [desc(x) for x in t]
    if isinstance(t, dict): return {k: desc(v) for k, v in sorted(t.items())}
    return t
out = {}
def run(name, fn):
    try: out[name] = desc(fn())
    except Exception as e: out[name] = {"error": type(e).__name__, "msg": str(e)[:70]}
emb = seeded(I.create_embedding_matrix(cols, 0.0001, False, device="cpu"))
lin = seeded(I.create_embedding_matrix(cols, 0.0001, True, device="cpu"))
run("emb_keys", lambda: {k: list(v.weight.shape) for k, v in emb.items()})
run("lin_keys", lambda: {k: list(v.weight.shape) for k, v in lin.items()})
sparse_cols = [c for c in cols if isinstance(c, SparseFeat) and not isinstance(c, VarLenSparseFeat)]
varlen_cols = [c for c in cols if isinstance(c, VarLenSparseFeat)]
run("embedding_lookup", lambda: I.embedding_lookup(Xt, emb, fi, sparse_cols))
run("embedding_lookup_list", lambda: I.embedding_lookup(Xt, emb, fi, sparse_cols, return_feat_list=("i",), to_list=True))
run("varlen_lookup", lambda: I.varlen_embedding_lookup(Xt, emb, fi, varlen_cols))
run("varlen_pooling", lambda: I.get_varlen_pooling_list(I.varlen_embedding_lookup(Xt, emb, fi, varlen_cols), Xt, fi, varlen_cols, "cpu"))
run("dense_input", lambda: I.get_dense_input(Xt, fi, cols))
run("maxlen_lookup", lambda: I.maxlen_lookup(Xt, fi, ["hl"]))
run("maxlen_lookup_bad", lambda: I.maxlen_lookup(Xt, fi, []))
run("combined", lambda: I.combined_dnn_input([torch.ones(B, 1, 4), torch.ones(B, 1, 4) * 2], [torch.ones(B, 2) * 3]))
run("combined_sparse_only", lambda: I.combined_dnn_input([torch.ones(B, 1, 4)], []))
run("combined_dense_only", lambda: I.combined_dnn_input([], [torch.ones(B, 2)]))
run("combined_none", lambda: I.combined_dnn_input([], []))
m = seeded(DeepFM(cols, cols, dnn_hidden_units=(8,), device="cpu"))
run("input_from_feature_columns", lambda: m.input_from_feature_columns(Xt, cols, m.embedding_dict))
run("input_from_feature_columns_nodense", lambda: m.input_from_feature_columns(Xt, cols, m.embedding_dict, support_dense=False))
run("linear_model", lambda: m.linear_model(Xt))
run("forward", lambda: m(Xt))
print("JSON" + json.dumps(out, sort_keys=True, default=str))
