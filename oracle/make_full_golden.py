"""Reference-pinned parity at BASELINE.json's FULL configurations (configs 2-4): runs the REAL reference
(shenweichen/DeepCTR-Torch, torch-CPU fp32, imported from /root/reference behind the TensorFlow stub of make_golden.py)
at 26 sparse x 1M-row vocabularies, 13 dense, embedding_dim 16, batch 4096 for DeepFM (256,128), xDeepFM (CIN [128,128]
split_half, dnn (256,256)) and FiBiNET ('interaction', 26 fields, dnn (128,128)), and stores what it computes:

    logit (pre-bias), y_pred, BCE(sum) loss, every dense gradient, the gradient of every TOUCHED table row, and the
    parameters after one reference train step (basemodel.py:242-262) under torch.optim.SGD and under
    torch.optim.Adagrad (accumulators preset to a positive constant: from zero the first Adagrad step is lr * sign(g)
    and would not see the gradient's magnitude).

Inputs and parameters are NOT stored: tests/fullsize_data.py generates them from an integer hash on both sides (the
tables' untouched rows cannot affect any output).  Tensors above 300k elements (FiBiNET's first tower layer) are stored as
a strided sample + four random projections, deep-table rows as every 16th touched row + projections over all of them.

    python oracle/make_full_golden.py [deepfm xdeepfm fibinet]       # ~1 min and ~12 GB of host memory per model
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullsize_data as FD  # noqa: E402
from make_golden import import_reference  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def build(ref, name):
    import deepctr_torch.inputs as ref_inputs
    import deepctr_torch.models as ref_models
    cols = FD.feature_columns(ref_inputs, FD.data_of(name))
    spec = FD.MODELS[name]
    return getattr(ref_models, spec["cls"])(cols, cols, l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0, seed=1024,
                                            device="cpu", **spec["kwargs"])


def set_params(model, touched, sparse):
    import torch
    with torch.no_grad():
        for k, p in model.state_dict().items():
            if "embedding_dict" in k:
                f = sparse.index(k.split(".")[-2])
                rows = touched[f]
                p[torch.from_numpy(rows)] = torch.from_numpy(FD.table_rows(k, rows, p.shape[1]))
            else:
                p.copy_(torch.from_numpy(FD.dense_param(k, tuple(p.shape))))


def store(out, prefix, d):
    for k, v in d.items():
        out[prefix + "/" + k] = v


def collect(out, tag, model, touched, get, sparse):
    """get(name, param) -> tensor to summarise (the gradient, or the updated parameter)"""
    for k, p in model.named_parameters():
        t = get(k, p).detach().numpy()
        if "embedding_dict" in k:
            f = sparse.index(k.split(".")[-2])
            rows = touched[f]
            if p.shape[1] == 1:
                out["%s/%s/all_touched" % (tag, k)] = t[rows, 0].astype(np.float32)
            else:
                store(out, "%s/%s" % (tag, k), FD.summarise_rows(rows, t[rows]))
        else:
            store(out, "%s/%s" % (tag, k), FD.summarise(t, FD.BIG if tag == "grad" else FD.BIG_STEP))


def run(ref, name):
    import torch
    import torch.nn.functional as F
    t0 = time.time()
    data = FD.data_of(name)
    sparse = FD.table_names(data)
    X, y = FD.inputs(data)
    touched = FD.touched_rows(X, data)
    model = build(ref, name)
    set_params(model, touched, sparse)
    start = {k: v.clone() for k, v in model.state_dict().items()}
    # checksums of what the hash generated HERE: the tests regenerate the same tensors and must find the same sums
    out = {"n_touched": np.array([len(r) for r in touched]),
           "check/X": np.array(float(X.astype(np.float64).sum())), "check/y": np.array(float(y.sum())),
           "check/table_C7": np.array(float(FD.table_rows("embedding_dict.C7.weight", touched[6], FD.DIM).astype(np.float64).sum())),
           "check/dnn0": np.array(float(FD.dense_param("dnn.linears.0.weight", (8, 429)).astype(np.float64).sum()))}
    cap = {}
    hook = model.out.register_forward_pre_hook(lambda m, inp: cap.__setitem__("logit", inp[0].detach().clone()))
    model.train()
    xt, yt = torch.from_numpy(X), torch.from_numpy(y)
    y_pred = model(xt).squeeze()
    hook.remove()
    loss = F.binary_cross_entropy(y_pred, yt, reduction="sum")
    model.zero_grad()
    loss.backward()
    out["logit"] = cap["logit"].numpy().reshape(-1).astype(np.float32)
    out["y_pred"] = y_pred.detach().numpy().astype(np.float32)
    out["loss"] = np.array(loss.item(), np.float64)
    collect(out, "grad", model, touched, lambda k, p: p.grad, sparse)
    model.zero_grad(set_to_none=True)
    # the same dense gradients with the reference evaluated in fp64: how far the reference's OWN fp32 gradient is from the
    # exact one (the CIN biases sum 65 536 terms; two fp32 summation orders differ by ~2e-5 relative there) -- the test
    # accepts max(2e-5 x max|g|, 2 x that gap)
    model.double()
    torch.set_default_dtype(torch.float64)        # (the reference creates a few tensors with the default dtype)
    try:
        loss64 = F.binary_cross_entropy(model(xt.double()).squeeze().double(), yt.double(), reduction="sum")
        loss64.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    out["loss64"] = np.array(loss64.item(), np.float64)
    for k, p in model.named_parameters():
        if "embedding_dict" not in k:
            store(out, "grad64/" + k, FD.summarise(p.grad.detach().numpy()))
    model.zero_grad(set_to_none=True)
    model.float()
    model.load_state_dict(start)
    for opt_name in ("sgd", "adagrad"):
        model.load_state_dict(start)
        model.compile(opt_name, "binary_crossentropy", metrics=[])
        if opt_name == "adagrad":
            for p in model.parameters():
                model.optim.state[p]["sum"].fill_(FD.ADAGRAD_SUM0)
        yp = model(xt).squeeze()                       # the reference's own step, basemodel.py:242-262
        model.optim.zero_grad()
        ls = model.loss_func(yp, yt, reduction="sum")
        total = ls + model.get_regularization_loss() + model.aux_loss
        total.backward()
        model.optim.step()
        out[opt_name + "_loss"] = np.array(ls.item(), np.float64)
        collect(out, opt_name, model, touched, lambda k, p: p, sparse)
        model.optim = None
        model.zero_grad(set_to_none=True)
    path = os.path.join(GOLDEN_DIR, "full", "%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-8s logit[min,max]=[%+.3f,%+.3f] loss=%.3f  -> %s (%.1f MB, %.0f s)" % (
        name, out["logit"].min(), out["logit"].max(), float(out["loss"]), os.path.relpath(path),
        os.path.getsize(path) / 2 ** 20, time.time() - t0))


if __name__ == "__main__":
    ref = import_reference()
    for n in (sys.argv[1:] or list(FD.MODELS)):
        run(ref, n)
