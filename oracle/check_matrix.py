"""Pin the numpy oracle on the reference's OWN test matrix (tests/models/*_test.py parameter lists as restated in
tests/test_gpu_reference_matrix.py): for every configuration, build the REAL reference model on torch-CPU, take its
freshly initialised state_dict, run its eval-mode forward on the same generated inputs and compare with
``np_oracle.Oracle.forward`` (fp64).  Runs only in the build container (needs /root/reference).  The reference's
parameters and predictions are stored in ``tests/golden/matrix/reference_matrix.npz`` (with the configurations as json):
the CPU tests re-check the oracle against them and compare ``state_dict`` keys / shapes, the GPU tests load the
parameters into the drop-in models and compare the HIP forward with the reference's predictions.

    python oracle/check_matrix.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def configs():
    """(model, seed, n_sparse, n_dense, kwargs, with_linear, include_length) per case of the matrix."""
    out = []
    for use_fm, hidden, ns, nd in [(True, (32,), 3, 3), (False, (32,), 3, 3), (False, (32,), 2, 2), (False, (32,), 1, 1),
                                   (True, (), 1, 1), (False, (), 2, 2), (True, (32,), 0, 3), (True, (32,), 3, 0),
                                   (False, (32,), 0, 3), (False, (32,), 3, 0)]:
        for lin in (True, False):
            out.append(("DeepFM", 1, ns, nd, dict(use_fm=use_fm, dnn_hidden_units=hidden), lin, False))
    out.append(("DeepFM", 2, 2, 2, dict(dnn_hidden_units=(32,)), True, True))
    for hidden, cin, sh, act, ns in [((), (), True, "linear", 1), ((8,), (), True, "linear", 1),
                                     ((), (8,), True, "linear", 2), ((8,), (8,), False, "relu", 2)]:
        out.append(("xDeepFM", 3, ns, ns, dict(dnn_hidden_units=hidden, cin_layer_size=cin, cin_split_half=sh,
                                               cin_activation=act), True, False))
    for bt in ("each", "interaction", "all"):
        out.append(("FiBiNET", 4, 3, 3, dict(bilinear_type=bt, dnn_hidden_units=[8, 8]), True, False))
    for cn, param in [(2, "vector"), (1, "matrix")]:
        out.append(("DCN", 5, 2, 2, dict(cross_num=cn, cross_parameterization=param, dnn_hidden_units=(32,)), True, False))
    out.append(("DCNMix", 6, 2, 2, dict(cross_num=1, dnn_hidden_units=(32,)), True, False))
    for ui, uo, kt, ns in [(True, True, "mat", 2), (True, False, "mat", 2), (False, True, "vec", 3),
                           (False, True, "num", 3), (False, False, "mat", 1)]:
        out.append(("PNN", 7, ns, ns, dict(dnn_hidden_units=[32, 32], use_inner=ui, use_outter=uo, kernel_type=kt),
                    False, False))
    for ns in (2, 1):
        out.append(("NFM", 8, ns, ns, dict(dnn_hidden_units=[32, 32]), True, False))
    out.append(("AFM", 9, 3, 0, dict(use_attention=True), True, False))
    for al, hidden, ns in [(1, (4,), 2), (2, (4, 4), 2), (1, (), 1), (1, (4,), 1)]:
        out.append(("AutoInt", 10, ns, ns, dict(att_layer_num=al, dnn_hidden_units=hidden), True, False))
    # towers OUTSIDE the MFMA kernels' envelope (they stay on PyTorch-ROCm; the numpy oracle does not state them, so
    # these are compared with the reference only): WDL's PReLU tower of WDL_test.py, BatchNorm, sigmoid
    for ns, nd in [(2, 0), (0, 2), (2, 2)]:
        out.append(("WDL", 11, ns, nd, dict(dnn_activation="prelu", dnn_hidden_units=[32, 32]), True, False))
    out.append(("DeepFM", 12, 2, 2, dict(dnn_hidden_units=(16, 8), dnn_use_bn=True), True, False))
    # (dnn_activation="dice" cannot run in the reference's own DeepFM: DNN builds Dice with dice_dim=3 for a 2-D input)
    out.append(("DeepFM", 12, 2, 2, dict(dnn_hidden_units=(16, 8), dnn_activation="sigmoid"), True, False))
    out.append(("DCN", 12, 2, 2, dict(dnn_hidden_units=(16,), dnn_use_bn=True, cross_num=2), True, False))
    # shapes OUTSIDE the interaction kernels' envelope (they run the same math as PyTorch-ROCm ops): CIN over more than
    # 32 fields, bilinear / AFM with wide embeddings or attention, a cross network wider than 2048 -- (.., emb) appended
    out.append(("xDeepFM", 13, 34, 2, dict(dnn_hidden_units=(8,), cin_layer_size=(6, 4), cin_split_half=True), True, False, 4))
    for bt in ("each", "interaction", "all"):
        out.append(("FiBiNET", 13, 3, 2, dict(bilinear_type=bt, dnn_hidden_units=[8]), True, False, 20))
    out.append(("AFM", 13, 3, 0, dict(use_attention=True, attention_factor=40), True, False, 8))
    out.append(("DCN", 13, 28, 2, dict(cross_num=2, cross_parameterization="vector", dnn_hidden_units=(8,)), True, False, 68))
    return out


def oracle_states(model, kw):
    return model != "WDL" and not kw.get("dnn_use_bn") and kw.get("dnn_activation", "relu") in ("relu", "linear")


def main():
    import torch
    import make_golden as mg
    from np_oracle import Oracle
    sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
    import matrix_data as tm                        # the generator + spec_of the GPU tests use
    ref = mg.import_reference()
    worst = 0.0
    store, meta = {}, []
    for cfg in configs():
        model, seed, ns, nd, kw, with_lin, inc_len = cfg[:7]
        emb = cfg[7] if len(cfg) > 7 else 4
        # (length_name + 'max' crashes in the reference itself on torch >= 1.2: sequence.py:66 subtracts a bool mask)
        x, y, cols = tm.make_data(seed, ns, nd, emb=emb, include_length=inc_len, seqs=("sum", "mean") if inc_len else
                                  ("sum", "mean", "max"), min_clean=tm.N if seed == 12 else 16)   # batch statistics: no -1e9 rows
        spec = tm.spec_of(model, cols if with_lin else [], cols, **kw)
        torch.manual_seed(0)
        m = mg.build_reference_model(ref, spec, l2=1e-5)
        m.eval()
        X = np.concatenate([np.asarray(x[name], np.float32).reshape(tm.N, -1) for name in m.feature_index], axis=1)
        cap = {}
        hook = m.out.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("logit", inp[0].detach().clone()))
        with torch.no_grad():
            want = m(torch.from_numpy(X)).numpy()
        hook.remove()
        params = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        okt = torch.from_numpy(tm.clean_rows(x, cols))
        # gradients of BCE(sum) over the rows with a defined value, in train mode (dropout is 0 here)
        yt = torch.from_numpy(np.asarray(y, np.float32))
        m.train()                                  # BatchNorm / Dice: batch statistics, as in a training step
        m.zero_grad()
        torch.nn.functional.binary_cross_entropy(m(torch.from_numpy(X)).squeeze(1)[okt], yt[okt], reduction="sum").backward()
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy() for k, p in m.named_parameters()}
        has_oracle = oracle_states(model, kw)
        lg, got = Oracle(spec, params, dtype=np.float64).forward(X) if has_oracle else (cap["logit"].numpy(), want)
        ok = tm.clean_rows(x, cols)          # rows with an all-padding 'max' field are ~1e9 noise in the reference itself
        err = float(np.max(np.abs(np.asarray(got).reshape(-1)[ok] - want.reshape(-1)[ok]))) if ok.any() else 0.0
        worst = max(worst, err)
        print("%-8s lin=%d %-90s max|d| = %.2e (%d rows)" % (model, with_lin, kw, err, int(ok.sum())))
        assert err <= 2e-6, (model, kw, err)
        lerr = float(np.max(np.abs(np.asarray(lg).reshape(-1)[ok] - cap["logit"].numpy().reshape(-1)[ok])))
        assert lerr <= 1e-5, (model, kw, lerr)
        i = len(meta)
        meta.append({"model": model, "seed": seed, "n_sparse": ns, "n_dense": nd, "kwargs": kw, "with_linear": with_lin,
                     "include_length": inc_len, "spec": spec})
        store["%d/X" % i], store["%d/y_pred" % i], store["%d/clean" % i] = X, want.reshape(-1, 1), ok
        store["%d/y" % i] = np.asarray(y, np.float32)
        store["%d/logit" % i] = cap["logit"].numpy().reshape(-1, 1)
        for k, v in params.items():
            store["%d/param/%s" % (i, k)] = v
        for k, v in grads.items():
            store["%d/grad/%s" % (i, k)] = v
        meta[-1]["oracle"] = has_oracle
        if kw.get("dnn_use_bn"):
            # Train-mode BatchNorm at initialisation (activations ~1e-4, divided by sqrt(var + 1e-5)) is ill-conditioned:
            # the SAME reference evaluated in fp64 says how far fp32 round-off alone moves every gradient.  Stored so
            # (the reference's in-place logit adds keep the wide / final logit in fp32; the tower and its BatchNorm --
            # where the conditioning problem lives -- run in fp64.)  Stored so
            # that the GPU test can bound its error by the reference's own fp32 uncertainty instead of skipping.
            import copy
            m64 = copy.deepcopy(m).double()
            m64.train()
            m64.zero_grad()
            torch.nn.functional.binary_cross_entropy(m64(torch.from_numpy(X).double()).squeeze(1)[okt].double(), yt[okt].double(),
                                                     reduction="sum").backward()
            for k, p in m64.named_parameters():
                store["%d/grad64/%s" % (i, k)] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        if not has_oracle:
            continue
        # the oracle's backward on the same rows
        o = Oracle(spec, params, dtype=np.float64)
        _, yp = o.forward(X)
        gl = (np.asarray(yp).reshape(-1, 1) - np.asarray(y, np.float64).reshape(-1, 1)) * ok.reshape(-1, 1)
        og = o.backward(gl)
        for k, v in grads.items():
            e = float(np.max(np.abs(np.asarray(og.get(k, np.zeros_like(v))).reshape(v.shape) - v))) if v.size else 0.0
            assert e <= 2e-5 * max(1.0, float(np.max(np.abs(v))) if v.size else 1.0), (model, kw, k, e)
    out = os.path.join(ROOT, "tests", "golden", "matrix")
    os.makedirs(out, exist_ok=True)
    store["configs"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out, "reference_matrix.npz"), **store)
    print("reference results stored for %d configurations (its test matrix + out-of-envelope towers); oracle == reference wherever it states the model (worst %.2e)" % (len(configs()), worst))


if __name__ == "__main__":
    main()
