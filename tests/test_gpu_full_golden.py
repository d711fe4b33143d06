"""GPU parity against the REAL reference at BASELINE.json's full configurations (configs 2-4): 26 sparse x 1M-row
vocabularies, 13 dense, embedding_dim 16, batch 4096 -- DeepFM (256,128), xDeepFM (CIN [128,128] split_half, dnn (256,256)),
FiBiNET ('interaction' over all 26 fields, dnn (128,128)).

tests/golden/full/*.npz hold what the reference computed in the build container (oracle/make_full_golden.py): logits,
predictions, loss, every dense gradient, the gradient of every touched table row, and the parameters after one train step
under SGD and under Adagrad.  Inputs and parameters are regenerated here from the same integer hash
(tests/fullsize_data.py).  Bars: logits 1e-5 absolute; gradients 2e-5 x max|gradient of the tensor| (+ the resolution of
a gradient read back from an SGD step, ulp(w) / lr); updated table rows 2e-5 absolute; updated dense parameters
2e-5 x max(1, |p|) + lr x the gradient's bar."""
import os

import numpy as np
import pytest
import torch

from np_oracle import optimizer_step

import fullsize_data as FD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden(name):
    z = np.load(os.path.join(GOLD, "full", "%s.npz" % name), allow_pickle=False)
    return {k: z[k] for k in z.files}


_CACHE = {}


def _data(name="deepfm"):
    data = FD.data_of(name)
    if data not in _CACHE:
        X, y = FD.inputs(data)
        _CACHE[data] = (X, y, FD.touched_rows(X, data))
    return _CACHE[data]


def _build(name):
    import deepctr_torch.inputs as I
    import deepctr_torch.models as M
    sparse = FD.table_names(FD.data_of(name))
    cols = FD.feature_columns(I, FD.data_of(name))
    spec = FD.MODELS[name]
    m = getattr(M, spec["cls"])(cols, cols, l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0, seed=1024, device=DEV,
                                **spec["kwargs"])
    X, y, touched = _data(name)
    with torch.no_grad():
        for k, p in m.state_dict().items():
            if "embedding_dict" in k:
                f = sparse.index(k.split(".")[-2])
                rows = touched[f]
                p[torch.from_numpy(rows).to(DEV)] = torch.from_numpy(FD.table_rows(k, rows, p.shape[1])).to(DEV)
            else:
                p.copy_(torch.from_numpy(FD.dense_param(k, tuple(p.shape))))
    return m


# Achieved error next to its bar for every tensor compared (round-3 verdict: the dense-gradient bar is widened for the
# tensors the MFMA kernels sum over 65 536 terms -- the margin must be visible): written to
# gpurun_out/full_golden_errors.json when the module's tests are done (copied to profiles/ per round).
_ERRS = []


def _note(tag, key, err, bar, n_split=None, n_used64=None, n=None):
    e = {"model": tag, "tensor": key, "max_abs_err": float(err), "bar": float(bar),
         "err_over_bar": float(err / bar) if bar > 0 else None}
    if n_split is not None:
        e.update(n_elements=int(n), n_reference_fp32_fp64_split=int(n_split), n_needed_the_fp64_side=int(n_used64))
    _ERRS.append(e)


def _either(got, ref32, ref64, bar):
    """The fp32-or-fp64 rule for dense tensors, bounded (round 6): an element may take the reference's fp64 value instead of
    its fp32 one ONLY where the fixture's own two evaluations are split by more than half the bar -- the elements of a weight
    row that a ReLU branch flip moved (see the comment at the call site).  Everywhere else the reference's fp32 value alone
    decides, so a kernel error cannot hide between the two.  Returns (max error under that rule, number of split elements,
    number of elements that needed the fp64 side); the counts go to full_golden_errors.json."""
    got, ref32, ref64 = (np.asarray(a, np.float64).reshape(-1) for a in (got, ref32, ref64))
    d32, d64 = np.abs(got - ref32), np.abs(got - ref64)
    split = np.abs(ref32 - ref64) > 0.5 * bar
    d = np.where(split, np.minimum(d32, d64), d32)
    used64 = split & (d32 > bar) & (d64 <= bar)
    return (float(d.max()) if d.size else 0.0), int(split.sum()), int(used64.sum())


@pytest.fixture(scope="module", autouse=True)
def _dump_errors():
    yield
    import json
    out = os.path.join(os.path.dirname(GOLD.rstrip(os.sep).rsplit(os.sep, 1)[0]), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        worst = sorted((e for e in _ERRS if e["err_over_bar"] is not None), key=lambda e: -e["err_over_bar"])[:12]
        with open(os.path.join(out, "full_golden_errors.json"), "w") as fh:
            json.dump({"n": len(_ERRS), "worst_by_err_over_bar": worst, "all": _ERRS}, fh, indent=1)
    except OSError:
        pass


def _check_dense(tag, key, got, g, tol_abs):
    """compare a dense tensor with its stored summary (all values, or strided sample + projections)"""
    got = np.asarray(got, np.float64)
    if key + "/all" in g:
        ref = g[key + "/all"].astype(np.float64)
        err = float(np.max(np.abs(got - ref))) if ref.size else 0.0
        _note(tag, key, err, tol_abs(ref))
        assert err <= tol_abs(ref), "%s %s: max|d| = %.3e (bar %.3e)" % (tag, key, err, tol_abs(ref))
        return
    flat = got.reshape(-1)
    ref = g[key + "/sample"].astype(np.float64)
    err = float(np.max(np.abs(flat[::FD.STRIDE] - ref)))
    bar = tol_abs(ref)
    _note(tag, key + " (sample)", err, bar)
    assert err <= bar, "%s %s (sample): max|d| = %.3e (bar %.3e)" % (tag, key, err, bar)
    for k in range(4):
        pr = float(np.dot(flat, FD.proj_weights(flat.size, k)))
        # a sum of n terms each within `bar`, random signs: sqrt(n) x bar is generous for a systematic error to show
        assert abs(pr - g[key + "/proj"][k]) <= bar * np.sqrt(flat.size) * 0.6 + 1e-9, "%s %s (projection %d)" % (tag, key, k)


def _check_rows(tag, key, rows, got_rows, g, tol_abs):
    """deep-table rows: every 16th touched row in full + projections over all touched rows"""
    got_rows = np.asarray(got_rows, np.float64)
    keep = rows % FD.ROW_KEEP == 0
    assert np.array_equal(rows[keep], g[key + "/rows"])
    ref = g[key + "/values"].astype(np.float64)
    bar = tol_abs(ref)
    err = float(np.max(np.abs(got_rows[keep] - ref)))
    _note(tag, key + " (rows)", err, bar)
    assert err <= bar, "%s %s (rows): max|d| = %.3e (bar %.3e)" % (tag, key, err, bar)
    idx = rows[:, None] * got_rows.shape[1] + np.arange(got_rows.shape[1], dtype=np.int64)[None, :]
    for k in range(4):
        pr = float(np.sum(got_rows * FD.sym(idx, 9100 + k, FD.SEED_P, 2.0).astype(np.float64)))
        assert abs(pr - g[key + "/proj"][k]) <= bar * np.sqrt(got_rows.size) * 0.6 + 1e-9, "%s %s (projection %d)" % (tag, key, k)


@pytest.mark.parametrize("name", list(FD.MODELS))
def test_full_size_logits_match_the_reference(name):
    g = _golden(name)
    m = _build(name)
    X, y, _ = _data(name)
    cap = {}
    hook = m.out.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("logit", inp[0].detach()))
    m.train()
    with torch.no_grad():
        y_pred = m(torch.from_numpy(X).to(DEV)).squeeze()
    hook.remove()
    torch.cuda.synchronize()
    m.model_plan().check_ids()
    logit = cap["logit"].reshape(-1).double().cpu().numpy()
    err = float(np.max(np.abs(logit - g["logit"].astype(np.float64))))
    _note(name, "logit", err, 1e-5)
    assert err <= 1e-5, "%s: logit max|d| = %.3e" % (name, err)
    assert float(np.max(np.abs(y_pred.double().cpu().numpy() - g["y_pred"]))) <= 5e-6
    loss = torch.nn.functional.binary_cross_entropy(y_pred, torch.from_numpy(y).to(DEV), reduction="sum").item()
    assert abs(loss - float(g["loss"])) <= 2e-5 * float(g["loss"])


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
@pytest.mark.parametrize("name", list(FD.MODELS))
def test_full_size_train_step_matches_the_reference(name, opt):
    """one train step through model._train_step (the fused step for DeepFM, autograd + the fused update kernels for the
    others) against the reference's step: the loss, every updated dense parameter, every touched row of every table;
    under SGD the gradients are read back from the step, g = (w0 - w1) / lr, and compared with the reference's autograd
    gradients as well"""
    g = _golden(name)
    m = _build(name)
    X, y, touched = _data(name)
    sparse = FD.table_names(FD.data_of(name))
    w0 = {k: v.detach().clone() for k, v in m.state_dict().items() if "embedding_dict" not in k}
    m.compile(opt, "binary_crossentropy", metrics=[])
    if opt == "adagrad":
        for grp in m.optim.param_groups:
            for p in grp["params"]:
                m.optim.state[p]["sum"].fill_(FD.ADAGRAD_SUM0)
    m.train()
    loss, _, _ = m._train_step(torch.from_numpy(X).to(DEV), torch.from_numpy(y).to(DEV))
    torch.cuda.synchronize()
    m.model_plan().check_ids()
    assert abs(loss.item() - float(g[opt + "_loss"])) <= 2e-5 * float(g[opt + "_loss"])
    sd = m.state_dict()
    lr = FD.LR_SGD
    for k, v in sd.items():
        key = "%s/%s" % (opt, k)
        if "embedding_dict" in k:
            f = sparse.index(k.split(".")[-2])
            rows = touched[f]
            got = v[torch.from_numpy(rows).to(DEV)].double().cpu().numpy()
            start = FD.table_rows(k, rows, v.shape[1]).astype(np.float64)
            if v.shape[1] == 1:
                ref = g[key + "/all_touched"].astype(np.float64)
                assert float(np.max(np.abs(got[:, 0] - ref))) <= 2e-5, "%s %s" % (name, key)
                if opt == "sgd":
                    gref = g["grad/%s/all_touched" % k].astype(np.float64)
                    gerr = float(np.max(np.abs((start[:, 0] - got[:, 0]) / lr - gref)))
                    assert gerr <= 2e-5 * float(np.max(np.abs(gref))) + 1e-6, "%s grad of %s: %.3e" % (name, k, gerr)
            else:
                _check_rows(name, key, rows, got, g, lambda ref: 2e-5)
                if opt == "sgd":
                    gmax = float(g["grad/%s/absmax" % k])
                    _check_rows(name, "grad/" + k, rows, (start - got) / lr, g, lambda ref: 2e-5 * gmax + 1e-6)
        else:
            # Dense parameters: every element must agree with the reference's fp32 result OR with its fp64 evaluation (both are
            # stored: grad/ and grad64/) within ONE relative bar, 2e-5 of the gradient's largest element.  Two references
            # because the tower is piecewise linear: where a first-layer pre-activation lies within ~1e-7 of zero, an fp32 and
            # an fp64 forward take different branches of the ReLU and that sample's whole contribution to a weight row
            # (~4e-4 of max |g| on FiBiNET's 10 413-wide layer) moves -- the reference's own two runs differ by that much; an
            # element on either branch is right.  The updated parameter is compared the same way: the reference's fp32
            # parameter, or the oracle's optimizer step (np_oracle.optimizer_step) on the fp64 gradient.  The gradient read
            # back from an SGD step, (w0 - w) / lr, is quantised to ulp(w) / lr (half an ulp from rounding w, up to one more
            # from the product): 1.5 ulp(w) / lr is the resolution of that measurement, added to its bar (FiBiNET's
            # bilinear matrices: max |g| ~ 3e-3 at |w| ~ 0.5 -- there the floor IS the bar; measured 1.1 ulp).
            # (Rounds 3-4 widened the bar itself by the reference's fp32-vs-fp64 gap, up to 1e-4.)
            got = v.double().cpu().numpy()
            gk, k64 = "grad/" + k, "grad64/" + k
            full = k64 + "/all" in g
            sfx = "/all" if full else "/sample"
            g64, g32 = np.asarray(g[k64 + sfx], np.float64), np.asarray(g[gk + sfx], np.float64)
            gmax = float(g[gk + "/absmax"]) if gk + "/absmax" in g else float(np.max(np.abs(g64)))
            pick = (lambda a: np.asarray(a, np.float64).reshape(g64.shape)) if full else \
                (lambda a: np.asarray(a, np.float64).reshape(-1)[::FD.STRIDE])
            w0k = pick(w0[k].double().cpu().numpy())
            lr_eff = lr if opt == "sgd" else FD.LR_ADAGRAD
            s0 = np.full_like(w0k, FD.ADAGRAD_SUM0) if opt == "adagrad" else None
            ref64, _ = optimizer_step(opt, w0k, g64, s0, lr_eff, 1e-10)
            # (the fixture may hold the updated parameter sampled where it holds the gradient in full, or the other way round)
            if key + sfx in g:
                ref32 = np.asarray(g[key + sfx], np.float64).reshape(ref64.shape)
            elif full:         # gradient in full, parameter sampled: compare on the samples
                ref32 = np.asarray(g[key + "/sample"], np.float64)
                ref64 = ref64.reshape(-1)[::FD.STRIDE]
                pick = lambda a: np.asarray(a, np.float64).reshape(-1)[::FD.STRIDE]      # noqa: E731
            else:              # gradient sampled, parameter in full
                ref32 = np.asarray(g[key + "/all"], np.float64).reshape(-1)[::FD.STRIDE]
            sens = 1.0 if opt == "sgd" else 1.0 / np.sqrt(FD.ADAGRAD_SUM0)      # d step / d g
            bar = 2e-5 * max(1.0, float(np.max(np.abs(ref64)))) + lr_eff * sens * 2e-5 * gmax
            gotk = pick(got)
            err, n_split, n_used64 = _either(gotk, ref32, ref64, bar)
            _note(name, key + " (fp32 | fp64 step)", err, bar, n_split, n_used64, gotk.size)
            assert err <= bar, "%s %s: max|d| = %.3e (bar %.3e)" % (name, key, err, bar)
            if key + "/proj" in g:      # the elements between the samples: four random projections of the whole tensor (fp32 reference)
                flat = got.reshape(-1)
                for q in range(4):
                    pr = float(np.dot(flat, FD.proj_weights(flat.size, q)))
                    assert abs(pr - g[key + "/proj"][q]) <= bar * np.sqrt(flat.size) * 0.6 + 1e-9, "%s %s (projection %d)" % (name, key, q)
            if opt == "sgd":
                wabs = max(float(w0[k].abs().max().item()), float(np.abs(got).max()))
                floor = 2.0 ** (np.floor(np.log2(max(wabs, 1e-30))) - 23) / lr     # ulp(w) / lr
                gotg = (pick(w0[k].double().cpu().numpy()) - gotk) / lr
                g32s, g64s = (g32.reshape(-1)[::FD.STRIDE], g64.reshape(-1)[::FD.STRIDE]) if gotg.shape != g64.shape else (g32, g64)
                gbar = 2e-5 * gmax + 1.5 * floor
                gerr, n_split, n_used64 = _either(gotg, g32s.reshape(gotg.shape), g64s.reshape(gotg.shape), gbar)
                _note(name, gk + " (fp32 | fp64)", gerr, gbar, n_split, n_used64, gotg.size)
                assert gerr <= gbar, "%s %s: max|d| = %.3e (bar %.3e, err / bar %.2f)" % (name, gk, gerr, gbar, gerr / gbar)
