"""Shared test helpers: golden fixtures, spec -> feature columns, oracle construction."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {"spec": json.loads(str(z["spec"])), "X": z["X"], "y": z["y"], "logit": z["logit"],
         "y_pred": z["y_pred"], "loss": float(z["loss"]), "params": {}, "grads": {}, "extra": {}}
    for k in z.files:
        if k.startswith("param/"):
            g["params"][k[6:]] = z[k]
        elif k.startswith("grad/"):
            g["grads"][k[5:]] = z[k]
        elif k not in ("spec", "X", "y", "logit", "y_pred", "loss"):
            g["extra"][k] = z[k]
    return g


def feature_columns(cols):
    """spec columns -> the drop-in package's SparseFeat / VarLenSparseFeat / DenseFeat."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    out = []
    for c in cols:
        if c["kind"] == "sparse":
            out.append(SparseFeat(c["name"], c["vocab"], c["dim"], embedding_name=c["embedding_name"]))
        elif c["kind"] == "dense":
            out.append(DenseFeat(c["name"], c["dimension"]))
        else:
            out.append(VarLenSparseFeat(SparseFeat(c["name"], c["vocab"], c["dim"], embedding_name=c["embedding_name"]),
                                        c["maxlen"], c["combiner"], c["length_name"]))
    return out


def build_model(spec, device, l2=0.0):
    """Instantiate the drop-in model class named by a golden spec."""
    import deepctr_torch.models as M
    lin, dnn = feature_columns(spec["linear_columns"]), feature_columns(spec["dnn_columns"])
    kw = dict(spec["kwargs"])
    if not hasattr(M, spec["model"]):
        import pytest
        pytest.skip("%s is not built yet" % spec["model"])
    cls = getattr(M, spec["model"])
    if spec["model"] == "PNN":
        return cls(dnn, l2_reg_embedding=l2, device=device, **kw)
    if spec["model"] == "AFM":
        return cls(lin, dnn, l2_reg_linear=l2, l2_reg_embedding=l2, l2_reg_att=l2, device=device, **kw)
    if spec["model"] == "AutoInt":
        return cls(lin, dnn, l2_reg_embedding=l2, device=device, **kw)
    if spec["model"] in ("DCN", "DCNMix"):
        return cls(lin, dnn, l2_reg_linear=l2, l2_reg_embedding=l2, l2_reg_cross=l2, device=device, **kw)
    return cls(lin, dnn, l2_reg_linear=l2, l2_reg_embedding=l2, device=device, **kw)


def load_params(model, params):
    import torch
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return model


def max_abs(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a.reshape(-1) - b.reshape(-1)))) if a.size else 0.0


def load_matrix():
    """tests/golden/matrix/reference_matrix.npz (oracle/check_matrix.py): the reference's freshly initialised parameters
    and eval-mode predictions on every configuration of its own model tests.  -> list of dicts."""
    z = np.load(os.path.join(GOLDEN_DIR, "matrix", "reference_matrix.npz"), allow_pickle=False)
    out = []
    for i, meta in enumerate(json.loads(str(z["configs"]))):
        pre, gpre, g64 = "%d/param/" % i, "%d/grad/" % i, "%d/grad64/" % i
        out.append(dict(meta, X=z["%d/X" % i], y=z["%d/y" % i], y_pred=z["%d/y_pred" % i], logit=z["%d/logit" % i], clean=z["%d/clean" % i],
                        params={k[len(pre):]: z[k] for k in z.files if k.startswith(pre)},
                        grads={k[len(gpre):]: z[k] for k in z.files if k.startswith(gpre)},
                        grads64={k[len(g64):]: z[k] for k in z.files if k.startswith(g64)}))
    return out


def matrix_id(c):
    kw = ",".join("%s=%s" % (k, str(v).replace(" ", "")) for k, v in sorted(c["kwargs"].items()))
    return "%s-%ds%dd-%s%s" % (c["model"], c["n_sparse"], c["n_dense"], kw, "" if c["with_linear"] else "-nolinear")
