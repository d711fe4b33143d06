"""Differential probe of the standalone layer classes (layers/interaction.py, core.py of the reference): same constructor
arguments, same seeded parameters and inputs, under the REAL reference and under the drop-in (on CPU tensors over
tests/mock_lib.py + tests/mock_ops.py) -- output shape / dtype / values, exception types, state_dict keys.  Build container
only.  Known, intended difference: FM on a 2-D input raises ValueError here (the reference: an incidental IndexError).

    python oracle/diff_layers.py          # runs both in subprocesses and reports
"""
import sys, json, os
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which == "both":
    import subprocess
    res = {}
    for w in ("ref", "mine"):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), w], capture_output=True, text=True, cwd="/tmp").stdout
        res[w] = json.loads([l for l in o.splitlines() if l.startswith("JSON")][-1][4:])
    bad = [k for k in res["ref"] if res["ref"][k] != res["mine"].get(k) and k != "FM_bad"]
    for k in res["ref"]:
        print("%-28s %s" % (k, "DIFFERENT  ref %s  mine %s" % (res["ref"][k], res["mine"].get(k)) if k in bad else "same"))
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
import deepctr_torch.layers as LY
from deepctr_torch.layers import interaction as IT
B, F, D = 6, 5, 8
def seeded(layer):
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, p in sorted(layer.state_dict().items()):
            if p.dtype.is_floating_point: p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    return layer
def x3(): return torch.randn(B, F, D, generator=torch.Generator().manual_seed(3))
def xl(): return [t for t in torch.randn(B, F, D, generator=torch.Generator().manual_seed(3)).split(1, dim=1)]
def x2(w): return torch.randn(B, w, generator=torch.Generator().manual_seed(4))
out = {}
def run(name, fn):
    try:
        y = fn()
        if torch.is_tensor(y):
            out[name] = {"shape": list(y.shape), "sum": round(float(y.double().sum()), 4), "absmax": round(float(y.abs().max()), 4),
                         "dtype": str(y.dtype)}
        else:
            out[name] = y
    except Exception as e:
        out[name] = {"error": type(e).__name__, "msg": str(e)[:60]}
run("FM", lambda: IT.FM()(x3()))
run("FM_bad", lambda: IT.FM()(x2(5)))
run("BiInteraction", lambda: IT.BiInteractionPooling()(x3()))
run("CIN", lambda: seeded(IT.CIN(F, (6, 4), "relu", True, 1e-5, 1024, "cpu"))(x3()))
run("CIN_nosplit_linear", lambda: seeded(IT.CIN(F, (6, 3), "linear", False, 1e-5, 1024, "cpu"))(x3()))
run("CIN_bad_dim", lambda: seeded(IT.CIN(F, (6, 4), device="cpu"))(x2(5)))
run("CIN_bad_cfg", lambda: IT.CIN(F, (5, 4), split_half=True, device="cpu"))
run("CIN_empty", lambda: IT.CIN(F, (), device="cpu"))
run("Inner", lambda: IT.InnerProductLayer(device="cpu")(xl()))
run("Inner_noreduce", lambda: IT.InnerProductLayer(reduce_sum=False, device="cpu")(xl()))
for t in ("all", "each", "interaction"):
    run("Bilinear_" + t, lambda t=t: seeded(IT.BilinearInteraction(F, D, t, 1024, "cpu"))(x3()))
run("Bilinear_bad", lambda: IT.BilinearInteraction(F, D, "nope", 1024, "cpu"))
run("Bilinear_baddim", lambda: seeded(IT.BilinearInteraction(F, D, "all", 1024, "cpu"))(x2(5)))
run("SENET", lambda: seeded(IT.SENETLayer(F, 2, 1024, "cpu"))(x3()))
run("SENET_bad", lambda: seeded(IT.SENETLayer(F, 2, 1024, "cpu"))(x2(5)))
run("CrossNet_vec", lambda: seeded(IT.CrossNet(12, 3, "vector", device="cpu"))(x2(12)))
run("CrossNet_mat", lambda: seeded(IT.CrossNet(12, 2, "matrix", device="cpu"))(x2(12)))
run("CrossNet_bad", lambda: IT.CrossNet(12, 2, "tensor", device="cpu"))
run("CrossNetMix", lambda: seeded(IT.CrossNetMix(12, 4, 3, 2, device="cpu"))(x2(12)))
run("AFM", lambda: seeded(IT.AFMLayer(D, 4, 0.0, 0, 1024, "cpu"))(xl()))
run("Interacting", lambda: seeded(IT.InteractingLayer(D, 2, True, False, 1024, "cpu"))(x3()))
run("Interacting_scaled_nores", lambda: seeded(IT.InteractingLayer(D, 4, False, True, 1024, "cpu"))(x3()))
run("Interacting_bad_heads", lambda: IT.InteractingLayer(D, 3, device="cpu"))
run("Interacting_bad_dim", lambda: seeded(IT.InteractingLayer(D, 2, device="cpu"))(x2(5)))
for t in ("mat", "vec", "num"):
    run("Outter_" + t, lambda t=t: seeded(IT.OutterProductLayer(F, D, t, 1024, "cpu"))(xl()))
# shapes outside the kernels' envelope (the drop-in switches to torch formulations; values must still be the reference's)
def big3(F_, D_): return torch.randn(3, F_, D_, generator=torch.Generator().manual_seed(5)) * 0.5
def bigl(F_, D_): return [t for t in big3(F_, D_).split(1, dim=1)]
run("CIN_34_fields", lambda: seeded(IT.CIN(34, (6, 4), "relu", True, 1e-5, 1024, "cpu"))(big3(34, 4)))
for t in ("all", "each", "interaction"):
    run("Bilinear_D20_" + t, lambda t=t: seeded(IT.BilinearInteraction(4, 20, t, 1024, "cpu"))(big3(4, 20)))
run("Bilinear_39x16", lambda: seeded(IT.BilinearInteraction(39, 16, "all", 1024, "cpu"))(big3(39, 16)))
run("SENET_40x32", lambda: seeded(IT.SENETLayer(40, 3, 1024, "cpu"))(big3(40, 32)))
run("AFM_A40", lambda: seeded(IT.AFMLayer(8, 40, 0.0, 0, 1024, "cpu"))(bigl(5, 8)))
run("AFM_56_fields", lambda: seeded(IT.AFMLayer(16, 8, 0.0, 0, 1024, "cpu"))(bigl(56, 16)))
run("CrossNet_2100", lambda: seeded(IT.CrossNet(2100, 2, "vector", device="cpu"))(torch.randn(3, 2100, generator=torch.Generator().manual_seed(6)) * 0.1))
run("Inner_45_noreduce", lambda: IT.InnerProductLayer(reduce_sum=False, device="cpu")(bigl(45, 16)))
run("Outter_vec_45", lambda: seeded(IT.OutterProductLayer(45, 16, "vec", 1024, "cpu"))(bigl(45, 16)))
run("Interacting_D40", lambda: seeded(IT.InteractingLayer(40, 2, True, False, 1024, "cpu"))(big3(5, 40)))
run("DNN", lambda: seeded(LY.DNN(12, (8, 4), device="cpu"))(x2(12)))
run("DNN_empty", lambda: LY.DNN(12, (), device="cpu"))
run("Prediction_binary", lambda: LY.PredictionLayer("binary")(x2(1)))
run("Prediction_bad", lambda: LY.PredictionLayer("ranking"))
run("keys", lambda: {n: sorted(getattr(IT, n)(*a).state_dict().keys()) for n, a in (("CIN", (F, (6, 4))), ("SENETLayer", (F, 2)), ("BilinearInteraction", (F, D, "each")),
     ("CrossNet", (12, 2)), ("CrossNetMix", (12, 4, 3, 2)), ("AFMLayer", (D, 4)), ("InteractingLayer", (D, 2)), ("OutterProductLayer", (F, D, "mat")))})
print("JSON" + json.dumps(out, sort_keys=True, default=str))
