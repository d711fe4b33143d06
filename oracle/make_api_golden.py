"""tests/golden/api/api_variants.npz: the user-facing call variants of ``compile / fit / evaluate / predict`` executed on
the REAL reference (torch-CPU) -- what a script written against DeepCTR-Torch may do besides the plain case:

    regress   task='regression', compile('adam', 'mse', ['mse']), reference-default L2, x as a dict
    listval   x as a LIST in feature_index order, validation_data=(list, y), compile('sgd', ..., ['logloss', 'accuracy'])
    shared    a VarLen history column sharing the item table (embedding_name), a DenseFeat of dimension 3, 'adagrad', ['auc']
    rmsprop   compile('rmsprop', ...): an optimizer outside the in-kernel set -> exact dense gradients + torch.optim
    instance  an optimizer INSTANCE (torch.optim.Adam(lr=0.01, weight_decay=1e-4)) and a loss CALLABLE
    l1        add_regularization_weight(..., l1=1e-3) on the tower weights and on one table after construction (basemodel.py:
              400-428): the regulariser is no longer L2-only, so neither the lazy nor the fused path may be taken
    split     WDL whose linear and deep sides use DIFFERENT column sets (one column only wide, one only deep, a max-pooled
              VarLen column on both) and per-column embedding sizes 6 / 3 / 5 / 2

Every run: the reference's initial state_dict, its History after 2 unshuffled epochs, evaluate() and predict().
Runs only in the build container (needs /root/reference):   python oracle/make_api_golden.py
                                                             python oracle/make_api_golden.py --schedule   (lr_schedule.npz)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

N, BATCH, EPOCHS = 160, 32, 2


def variants():
    base = mg.criteo_columns(4, 2, 10, 4)
    shared = [mg.sparse("item", 12, 4), mg.varlen("hist_item", 12, 4, 4, "mean", embedding_name="item"),
              mg.dense("d3", 3), mg.sparse("user", 7, 4)]
    split_dnn = [mg.sparse("a", 9, 6), mg.sparse("b", 7, 3), mg.varlen("c", 8, 5, 4, "max"), mg.dense("d2", 2)]
    split_lin = [mg.sparse("b", 7, 3), mg.dense("e", 1), mg.varlen("c", 8, 5, 4, "max"), mg.sparse("z", 5, 2)]
    return [
        dict(tag="l1", cols=base, kwargs=dict(dnn_hidden_units=(8,)), l2=1e-5, opt="adagrad", loss="binary_crossentropy",
             metrics=["binary_crossentropy"], y="binary", x="dict", val="split", l1=1e-3, gpu=False),
        dict(tag="split", model="WDL", cols=split_dnn, lin_cols=split_lin, kwargs=dict(dnn_hidden_units=(8,)), l2=0.0,
             opt="adagrad", loss="binary_crossentropy", metrics=["binary_crossentropy"], y="binary", x="dict",
             val="split", gpu=False),
        dict(tag="regress", cols=base, kwargs=dict(task="regression", dnn_hidden_units=(8,)), l2=1e-5, opt="adam",
             loss="mse", metrics=["mse"], y="real", x="dict", val="split"),
        dict(tag="listval", cols=base, kwargs=dict(dnn_hidden_units=(8, 4)), l2=1e-5, opt="sgd",
             loss="binary_crossentropy", metrics=["logloss", "accuracy"], y="binary", x="list", val="data"),
        dict(tag="shared", cols=shared, kwargs=dict(dnn_hidden_units=(8,)), l2=0.0, opt="adagrad",
             loss="binary_crossentropy", metrics=["auc"], y="binary", x="dict", val="split"),
        dict(tag="rmsprop", cols=base, kwargs=dict(dnn_hidden_units=(8,)), l2=1e-5, opt="rmsprop",
             loss="binary_crossentropy", metrics=["binary_crossentropy"], y="binary", x="dict", val="split"),
        dict(tag="instance", cols=base, kwargs=dict(dnn_hidden_units=(8,)), l2=0.0, opt="instance",
             loss="callable", metrics=["acc"], y="binary", x="dict", val="split"),
    ]


def model_input(spec, X, as_list):
    """dict name -> array (2-D for VarLen / multi-dim dense columns), or the list in feature_index order."""
    from np_oracle import build_input_features
    fi = build_input_features(spec["linear_columns"] + spec["dnn_columns"])
    d = {}
    for name, (lo, hi) in fi.items():
        d[name] = X[:, lo] if hi - lo == 1 else X[:, lo:hi]
    return [d[k] for k in fi] if as_list else d


def main():
    import torch
    import torch.nn.functional as F
    ref = mg.import_reference()
    store, meta = {}, []
    for v in variants():
        rng = np.random.default_rng(4242 + sum(map(ord, v["tag"])))
        spec = {"model": v.get("model", "DeepFM"), "linear_columns": v.get("lin_cols", v["cols"]),
                "dnn_columns": v["cols"], "kwargs": v["kwargs"]}
        torch.manual_seed(0)
        m = mg.build_reference_model(ref, spec, l2=v["l2"])
        mg.randomise(m, rng)
        X, y = mg.synth_inputs(spec, N, rng)
        if v["y"] == "real":
            y = rng.normal(0.3, 1.0, N).astype(np.float32)
        Xv, yv = mg.synth_inputs(spec, 48, rng)
        t = v["tag"]
        store[t + "/X"], store[t + "/y"], store[t + "/Xv"], store[t + "/yv"] = X, y, Xv, yv
        for k, p in m.state_dict().items():
            store[t + "/param/" + k] = p.detach().numpy().copy()
        if v.get("l1"):           # user-added L1 terms: on the tower weights and on one embedding table
            m.add_regularization_weight(filter(lambda kv: "weight" in kv[0], m.dnn.named_parameters()), l1=v["l1"])
            m.add_regularization_weight(m.embedding_dict["C1"].weight, l1=v["l1"])
        opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-4) if v["opt"] == "instance" else v["opt"]
        loss = F.binary_cross_entropy if v["loss"] == "callable" else v["loss"]
        m.compile(opt, loss, metrics=v["metrics"])
        xin = model_input(spec, X, v["x"] == "list")
        torch.manual_seed(5)
        if v["val"] == "data":
            hist = m.fit(xin, y, batch_size=BATCH, epochs=EPOCHS, verbose=2, shuffle=False,
                         validation_data=(model_input(spec, Xv, True), yv))
        else:
            hist = m.fit(xin, y, batch_size=BATCH, epochs=EPOCHS, verbose=2, shuffle=False, validation_split=0.2)
        for k, val in hist.history.items():
            store["%s/hist/%s" % (t, k)] = np.asarray(val, np.float64)
        ev = m.evaluate(model_input(spec, Xv, v["x"] == "list"), yv, batch_size=20)
        for k, val in ev.items():
            store["%s/eval/%s" % (t, k)] = np.asarray(val, np.float64)
        store[t + "/pred"] = m.predict(xin, batch_size=50)
        for k, p in m.state_dict().items():
            store[t + "/final/" + k] = p.detach().numpy().copy()
        meta.append(dict(v, spec=spec))
        print(t, {k: np.round(val, 4).tolist() for k, val in hist.history.items()}, ev)
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "api")
    os.makedirs(out, exist_ok=True)
    store["variants"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out, "api_variants.npz"), **store)


if __name__ == "__main__" and "--schedule" not in sys.argv:
    main()


# ---- learning-rate schedules on model.optim (tests/golden/api/lr_schedule.npz) ------------------------------------------
SCHED_RUNS = (("adagrad0", "adagrad", 0.0), ("adam", "adam", 1e-5), ("sgd", "sgd", 1e-3), ("adagrad", "adagrad", 1e-3))
SCHED_STEPS, SCHED_AT, SCHED_GAMMA = 9, (3, 6), 0.5


def lr_schedule():
    """9 reference training steps (basemodel.py:242-262) of DeepFM with ``param_groups[*]['lr'] *= 0.5`` after steps 3
    and 6 -- what ``torch.optim.lr_scheduler.StepLR(model.optim, 3, 0.5)`` stepped from a callback does.  Small batches
    over 20-row vocabularies: rows wait several steps between touches, so the lazily replayed rows cross the boundaries."""
    import torch
    ref = mg.import_reference()
    cols = mg.criteo_columns(8, 3, 20, 8)
    spec = {"model": "DeepFM", "linear_columns": cols, "dnn_columns": cols, "kwargs": {"dnn_hidden_units": (16, 8)}}
    rng = np.random.default_rng(99)
    torch.manual_seed(0)
    m0 = mg.build_reference_model(ref, spec)
    mg.randomise(m0, rng)
    start = {k: v.clone() for k, v in m0.state_dict().items()}
    Xs, ys = zip(*[mg.synth_inputs(spec, 24, rng) for _ in range(SCHED_STEPS)])
    store = {"spec": np.array(json.dumps(spec)), "X": np.stack(Xs), "y": np.stack(ys)}
    for k, v in start.items():
        store["param/" + k] = v.numpy().copy()
    for tag, opt, l2 in SCHED_RUNS:
        torch.manual_seed(0)
        m = mg.build_reference_model(ref, spec, l2=l2)
        m.load_state_dict(start)
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        tot = []
        for i, (Xb, yb) in enumerate(zip(Xs, ys)):
            if i in SCHED_AT:
                for grp in m.optim.param_groups:
                    grp["lr"] *= SCHED_GAMMA
            yp = m(torch.from_numpy(Xb)).squeeze()
            m.optim.zero_grad()
            total = m.loss_func(yp, torch.from_numpy(yb), reduction="sum") + m.get_regularization_loss() + m.aux_loss
            total.backward()
            m.optim.step()
            tot.append(total.item())
        store[tag + "/total"] = np.asarray(tot, np.float64)
        for k, v in m.state_dict().items():
            store["%s/final/%s" % (tag, k)] = v.detach().numpy().copy()
        print(tag, np.round(tot, 4).tolist())
    # frozen tables (pretrained embeddings): requires_grad_(False) on one deep and one wide table before compile();
    # autograd hands the optimizer no gradient for them, their L2 term still counts in the logged loss
    for tag, opt, l2 in (("freeze_adagrad", "adagrad", 0.0), ("freeze_adam", "adam", 1e-5)):
        torch.manual_seed(0)
        m = mg.build_reference_model(ref, spec, l2=l2)
        m.load_state_dict(start)
        m.embedding_dict["C1"].weight.requires_grad_(False)
        m.linear_model.embedding_dict["C2"].weight.requires_grad_(False)
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        tot = []
        for Xb, yb in list(zip(Xs, ys))[:4]:
            yp = m(torch.from_numpy(Xb)).squeeze()
            m.optim.zero_grad()
            total = m.loss_func(yp, torch.from_numpy(yb), reduction="sum") + m.get_regularization_loss() + m.aux_loss
            total.backward()
            m.optim.step()
            tot.append(total.item())
        store[tag + "/total"] = np.asarray(tot, np.float64)
        for k, v in m.state_dict().items():
            store["%s/final/%s" % (tag, k)] = v.detach().numpy().copy()
        print(tag, np.round(tot, 4).tolist())
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "api")
    np.savez_compressed(os.path.join(out, "lr_schedule.npz"), **store)


if __name__ == "__main__" and "--schedule" in sys.argv:
    lr_schedule()
