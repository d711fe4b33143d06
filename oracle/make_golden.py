"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE (shenweichen/DeepCTR-Torch v0.2.9, torch-CPU fp32).

Runs only in the build container, where ``/root/reference`` is mounted (the GPU box does not have it --
that is why the vectors are committed).  The reference hard-imports TensorFlow's Keras callbacks
(``models/basemodel.py:22-25``, ``callbacks.py:2-4``); TensorFlow is not installed, so a minimal stub is
injected first.  Nothing from the reference is copied: it is imported, driven with
seeded inputs, and its outputs / autograd gradients / optimizer trajectories are stored.

    python oracle/make_golden.py            # rewrites every fixture (deterministic)

Each fixture holds: spec (json), X, y, param/<state_dict key>, logit (pre-bias, pre-sigmoid), y_pred,
loss (BCE, reduction='sum'), grad/<key> for every parameter, and for the DeepFM cases the parameters after
3 reference training steps with torch.optim.SGD / Adagrad defaults (``sgd3/<key>``, ``adagrad3/<key>``).
"""
import importlib.machinery as machinery
import json
import os
import sys
import types

import numpy as np

REFERENCE = os.environ.get("DCTR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference():
    """Import ``deepctr_torch`` from the reference tree behind a TensorFlow stub."""

    class Callback(object):
        def set_model(self, model):
            self.model = model

        def on_train_begin(self, logs=None):
            pass

        def on_train_end(self, logs=None):
            pass

        def on_epoch_begin(self, epoch, logs=None):
            pass

        def on_epoch_end(self, epoch, logs=None):
            pass

    class History(Callback):
        def on_train_begin(self, logs=None):
            self.epoch, self.history = [], {}

        def on_epoch_end(self, epoch, logs=None):
            self.epoch.append(epoch)
            for k, v in (logs or {}).items():
                self.history.setdefault(k, []).append(v)

    class CallbackList(object):
        def __init__(self, callbacks=None):
            self.callbacks = list(callbacks or [])

        def set_model(self, model):
            self.model = model
            for c in self.callbacks:
                c.set_model(model)

        def __getattr__(self, name):
            if name.startswith("on_"):
                return lambda *a, **k: [getattr(c, name)(*a, **k) for c in self.callbacks]
            raise AttributeError(name)

    class EarlyStopping(Callback):
        pass

    class ModelCheckpoint(Callback):
        pass

    for name in ["tensorflow", "tensorflow.python", "tensorflow.python.keras", "tensorflow.python.keras.callbacks"]:
        mod = types.ModuleType(name)
        mod.__spec__ = machinery.ModuleSpec(name, None, is_package=True)  # torch._dynamo calls find_spec
        mod.__path__ = []
        sys.modules[name] = mod
    cb = sys.modules["tensorflow.python.keras.callbacks"]
    cb.CallbackList, cb.History, cb.EarlyStopping, cb.ModelCheckpoint = CallbackList, History, EarlyStopping, \
        ModelCheckpoint
    for k in [k for k in sys.modules if k == "deepctr_torch" or k.startswith("deepctr_torch.")]:
        del sys.modules[k]  # never mix with the drop-in package of the same name
    sys.path.insert(0, REFERENCE)
    import deepctr_torch  # noqa: F401  (the reference)
    assert os.path.realpath(deepctr_torch.__file__).startswith(os.path.realpath(REFERENCE)), deepctr_torch.__file__
    return deepctr_torch


# --------------------------------------------------------------------------------------------------
# specs
# --------------------------------------------------------------------------------------------------
def sparse(name, vocab, dim, embedding_name=None):
    return {"kind": "sparse", "name": name, "vocab": vocab, "dim": dim, "embedding_name": embedding_name or name}


def varlen(name, vocab, dim, maxlen, combiner, length_name=None, embedding_name=None):
    return {"kind": "varlen", "name": name, "vocab": vocab, "dim": dim, "maxlen": maxlen, "combiner": combiner,
            "length_name": length_name, "embedding_name": embedding_name or name}


def dense(name, dimension=1):
    return {"kind": "dense", "name": name, "dimension": dimension}


def criteo_columns(n_sparse=26, n_dense=13, vocab=48, dim=16):
    return [sparse("C%d" % (i + 1), vocab + i, dim) for i in range(n_sparse)] + \
        [dense("I%d" % (i + 1)) for i in range(n_dense)]


def mixed_columns(dim=4):
    return [sparse("user", 11, dim), sparse("item", 9, dim), sparse("cate", 6, dim),
            dense("price"), dense("ctx", 3),
            varlen("hist_sum", 9, dim, 4, "sum", embedding_name="item"),     # shares the `item` table
            varlen("tags_mean", 7, dim, 5, "mean"),
            varlen("kw_max", 8, dim, 3, "max"),
            varlen("seq_len_mean", 6, dim, 4, "mean", length_name="seq_len_mean_length")]


CASES = []


def case(name, model, lin, dnn, batch=64, seed=0, steps=False, lazy=False, fit=False, **kwargs):
    CASES.append({"name": name, "batch": batch, "seed": seed, "steps": steps, "lazy": lazy, "fit": fit,
                  "spec": {"model": model, "linear_columns": lin, "dnn_columns": dnn, "kwargs": kwargs}})


_c = criteo_columns()
case("deepfm_criteo", "DeepFM", _c, _c, batch=96, steps=True, dnn_hidden_units=(256, 128))
_m = mixed_columns()
case("deepfm_mixed", "DeepFM", _m, _m, batch=64, steps=True, dnn_hidden_units=(32, 16))
case("deepfm_nolinear_nofm", "DeepFM", [], _m, batch=33, use_fm=False, dnn_hidden_units=(16,))
case("deepfm_dense_only", "DeepFM", [dense("a"), dense("b", 2)], [dense("a"), dense("b", 2)], batch=17,
     dnn_hidden_units=(8,))
case("deepfm_fm_only", "DeepFM", _m[:3], _m[:3], batch=20, dnn_hidden_units=())
case("xdeepfm_criteo", "xDeepFM", _c, _c, batch=48, dnn_hidden_units=(64, 64), cin_layer_size=(128, 128),
     cin_split_half=True)
_x = criteo_columns(6, 2, 20, 8)
case("xdeepfm_nosplit", "xDeepFM", _x, _x, batch=40, dnn_hidden_units=(16,), cin_layer_size=(10, 7, 5),
     cin_split_half=False)
case("xdeepfm_linear_act", "xDeepFM", _x, _x, batch=40, dnn_hidden_units=(), cin_layer_size=(8, 6),
     cin_split_half=True, cin_activation="linear")
_f = criteo_columns(10, 13, 30, 16)
case("fibinet_interaction", "FiBiNET", _f, _f, batch=48, dnn_hidden_units=(32, 16), bilinear_type="interaction")
_f2 = criteo_columns(5, 2, 12, 8)
case("fibinet_each", "FiBiNET", _f2, _f2, batch=32, dnn_hidden_units=(16,), bilinear_type="each", reduction_ratio=2)
case("fibinet_all", "FiBiNET", _f2, _f2, batch=32, dnn_hidden_units=(16,), bilinear_type="all", reduction_ratio=1)
_d = criteo_columns(8, 5, 25, 8)
case("dcn_vector", "DCN", _d, _d, batch=48, dnn_hidden_units=(32, 16), cross_num=2, cross_parameterization="vector")
case("dcn_matrix", "DCN", _d, _d, batch=48, dnn_hidden_units=(32, 16), cross_num=3, cross_parameterization="matrix")
# (DCN with dnn_hidden_units=() cannot be built in the reference: dcn.py:54 constructs DNN unconditionally)
case("pnn_inner", "PNN", [], _d, batch=48, dnn_hidden_units=(32, 16), use_inner=True, use_outter=False)
_pm = mixed_columns(8)
case("pnn_inner_varlen", "PNN", [], [c for c in _pm if c["kind"] != "dense"], batch=29, dnn_hidden_units=(16,),
     use_inner=True, use_outter=False)


_n = criteo_columns(9, 4, 22, 8)
case("nfm_criteo", "NFM", _n, _n, batch=40, steps=True, dnn_hidden_units=(32, 16))
case("nfm_sparse_only", "NFM", [c for c in _n if c["kind"] == "sparse"], [c for c in _n if c["kind"] == "sparse"],
     batch=24, dnn_hidden_units=(16,))

_a = criteo_columns(9, 3, 18, 8)
_as = [c for c in _a if c["kind"] == "sparse"]        # AFM rejects dense features on the deep side (afm.py:62-63)
case("afm_criteo", "AFM", _a, _as, batch=40, steps=True, attention_factor=8)
case("afm_wide_factor", "AFM", criteo_columns(5, 0, 12, 4), criteo_columns(5, 0, 12, 4), batch=24, attention_factor=5)
case("afm_no_attention", "AFM", _a, _as, batch=24, use_attention=False)

case("wdl_criteo", "WDL", _a, _a, batch=40, steps=True, dnn_hidden_units=(32, 16))
case("wdl_mixed", "WDL", _m, _m, batch=33, dnn_hidden_units=(16,))

_ai = criteo_columns(7, 3, 20, 8)
case("autoint_deep", "AutoInt", _ai, _ai, batch=40, steps=True, att_layer_num=2, att_head_num=2, dnn_hidden_units=(32, 16))
case("autoint_only_att", "AutoInt", _ai, _ai, batch=24, att_layer_num=3, att_head_num=4, att_res=False, dnn_hidden_units=())
case("dcnmix_deep", "DCNMix", _d, _d, batch=48, steps=True, dnn_hidden_units=(32, 16), cross_num=2, low_rank=8,
     num_experts=3)
# (DCNMix with dnn_hidden_units=() cannot be built in the reference either: dcnmix.py:56 constructs DNN unconditionally)
case("dcnmix_wide_experts", "DCNMix", _d, _d, batch=24, dnn_hidden_units=(8,), cross_num=3, low_rank=4, num_experts=2)

case("pnn_outer_mat", "PNN", [], _d, batch=40, steps=True, dnn_hidden_units=(32, 16), use_inner=True, use_outter=True,
     kernel_type="mat")
case("pnn_outer_vec", "PNN", [], _d, batch=24, dnn_hidden_units=(16,), use_inner=False, use_outter=True, kernel_type="vec")
case("pnn_outer_num", "PNN", [], _d, batch=24, dnn_hidden_units=(16,), use_inner=True, use_outter=True, kernel_type="num")

# regularised / Adam trajectories (the reference's DEFAULT kind of training: l2 > 0 on every table, basemodel.py:412-428,
# and torch.optim.Adam, basemodel.py:447-461): small batches over small vocabularies, so that most rows are NOT touched
# by a given step and are touched again a few steps later -- what the exact lazy update (csrc/lazy.hip) must replay
_l = criteo_columns(8, 3, 20, 8)
case("lazy_deepfm", "DeepFM", _l, _l, batch=24, lazy=True, dnn_hidden_units=(16, 8))
_ld = criteo_columns(6, 2, 16, 8)
case("lazy_dcn", "DCN", _ld, _ld, batch=24, lazy=True, dnn_hidden_units=(16,), cross_num=2)
LAZY_STEPS, LAZY_L2 = 8, 1e-3

# model.fit() itself (basemodel.py:137-309): History contents and predict() after 3 epochs over 300 rows, batch 64 with
# a 25 % validation split (225 train rows -> batches 64, 64, 64, 33), metrics binary_crossentropy + auc.  Three runs:
#   plain     l2 = 0, adagrad, shuffle=False
#   shuffled  l2 = 0, adagrad, shuffle=True after torch.manual_seed(FIT_SEED): the DataLoader's permutations
#   default   the reference's default kwargs (l2 = 1e-5 on tables and Linear) with adam, shuffle=True, same seed
_ft = criteo_columns(8, 3, 20, 8)
case("fit_deepfm", "DeepFM", _ft, _ft, batch=64, fit=True, dnn_hidden_units=(16, 8))
# a model outside the fused train step (autograd + torch.optim around the kernels) and one with an MFMA interaction
case("fit_dcn", "DCN", _ft, _ft, batch=64, fit=True, dnn_hidden_units=(16,), cross_num=2)
case("fit_xdeepfm", "xDeepFM", _ft, _ft, batch=64, fit=True, dnn_hidden_units=(16,), cin_layer_size=(8, 6))
FIT_ROWS, FIT_EPOCHS, FIT_SPLIT, FIT_SEED = 300, 3, 0.25, 777
FIT_RUNS = (("plain", "adagrad", 0.0, False), ("shuffled", "adagrad", 0.0, True), ("default", "adam", 1e-5, True))


# --------------------------------------------------------------------------------------------------
def ref_columns(ref_inputs, cols):
    out = []
    for c in cols:
        if c["kind"] == "sparse":
            out.append(ref_inputs.SparseFeat(c["name"], c["vocab"], c["dim"], embedding_name=c["embedding_name"]))
        elif c["kind"] == "dense":
            out.append(ref_inputs.DenseFeat(c["name"], c["dimension"]))
        else:
            sf = ref_inputs.SparseFeat(c["name"], c["vocab"], c["dim"], embedding_name=c["embedding_name"])
            out.append(ref_inputs.VarLenSparseFeat(sf, c["maxlen"], c["combiner"], c["length_name"]))
    return out


def synth_inputs(spec, batch, rng):
    """X [B, n_cols] float32 in build_input_features order + labels."""
    sys.path.insert(0, HERE)
    from np_oracle import build_input_features
    fi = build_input_features(spec["linear_columns"] + spec["dnn_columns"])
    width = max(hi for _, hi in fi.values())
    X = np.zeros((batch, width), np.float32)
    seen = set()
    for c in spec["linear_columns"] + spec["dnn_columns"]:
        if c["name"] in seen:
            continue
        seen.add(c["name"])
        lo, hi = fi[c["name"]]
        if c["kind"] == "sparse":
            X[:, lo] = rng.integers(0, c["vocab"], batch)
            X[:batch // 8, lo] = X[0, lo]                      # force duplicate ids inside the batch
        elif c["kind"] == "dense":
            X[:, lo:hi] = rng.random((batch, hi - lo), dtype=np.float32)
        else:
            T = hi - lo
            # empty sequences (all padding) are included, except under 'max' pooling where the reference
            # turns an empty sequence into an embedding of -1e9 (sequence.py:65-68) and the logit with it
            lens = rng.integers(1 if c["combiner"] == "max" else 0, T + 1, batch)
            ids = rng.integers(1, c["vocab"], (batch, T))
            ids[np.arange(T)[None, :] >= lens[:, None]] = 0     # 0 = padding id
            X[:, lo:hi] = ids
            if c.get("length_name"):
                X[:, fi[c["length_name"]][0]] = lens
    y = rng.integers(0, 2, batch).astype(np.float32)
    return X, y


def randomise(model, rng):
    """'Trained-like' weights so that |logit| reaches O(1)."""
    import torch
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "embedding_dict" in k:
                p.copy_(torch.from_numpy(rng.normal(0, 0.15, tuple(p.shape)).astype(np.float32)))
            elif k.endswith("bias"):
                p.copy_(torch.from_numpy(rng.normal(0, 0.05, tuple(p.shape)).astype(np.float32)))
            else:
                fan_in = p.shape[1] if p.dim() >= 2 else p.shape[0]
                if k.startswith("crossnet.kernels"):
                    fan_in = p.shape[1]
                std = 1.2 / np.sqrt(max(1, fan_in))
                if k in ("linear_model.weight",):
                    std = 0.3
                p.copy_(torch.from_numpy(rng.normal(0, std, tuple(p.shape)).astype(np.float32)))


def build_reference_model(ref, spec, l2=0.0):
    import deepctr_torch.inputs as ref_inputs
    import deepctr_torch.models as ref_models
    lin, dnn = ref_columns(ref_inputs, spec["linear_columns"]), ref_columns(ref_inputs, spec["dnn_columns"])
    kw = dict(spec["kwargs"])
    cls = getattr(ref_models, spec["model"])
    if spec["model"] == "PNN":
        return cls(dnn, l2_reg_embedding=l2, device="cpu", **kw)
    if spec["model"] == "AFM":
        return cls(lin, dnn, l2_reg_linear=l2, l2_reg_embedding=l2, l2_reg_att=l2, device="cpu", **kw)
    if spec["model"] == "AutoInt":
        return cls(lin, dnn, l2_reg_embedding=l2, device="cpu", **kw)
    if spec["model"] in ("DCN", "DCNMix"):
        return cls(lin, dnn, l2_reg_linear=l2, l2_reg_embedding=l2, l2_reg_cross=l2, device="cpu", **kw)
    return cls(lin, dnn, l2_reg_linear=l2, l2_reg_embedding=l2, device="cpu", **kw)


ADAGRAD_SUM0 = 0.05


def run_case(ref, case_):
    import torch
    import torch.nn.functional as F
    spec, batch = case_["spec"], case_["batch"]
    rng = np.random.default_rng(1000 + case_["seed"] + sum(map(ord, case_["name"])))
    torch.manual_seed(case_["seed"])
    model = build_reference_model(ref, spec)
    randomise(model, rng)
    X, y = synth_inputs(spec, batch, rng)
    out = {"spec": np.array(json.dumps(spec)), "X": X, "y": y}
    for k, v in model.state_dict().items():
        out["param/" + k] = v.detach().numpy().copy()

    captured = {}
    hook = model.out.register_forward_pre_hook(lambda m, inp: captured.__setitem__("logit", inp[0].detach().clone()))
    model.train()
    xt, yt = torch.from_numpy(X), torch.from_numpy(y)
    y_pred = model(xt).squeeze()
    hook.remove()
    loss = F.binary_cross_entropy(y_pred, yt, reduction="sum")
    model.zero_grad()
    loss.backward()
    out["logit"] = captured["logit"].numpy().reshape(-1, 1)
    out["y_pred"] = y_pred.detach().numpy().reshape(-1, 1)
    out["loss"] = np.array(loss.item(), np.float64)
    for k, p in model.named_parameters():
        out["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()

    if case_["steps"]:
        Xs, ys = zip(*[synth_inputs(spec, batch, rng) for _ in range(3)])
        out["X_steps"], out["y_steps"] = np.stack(Xs), np.stack(ys)
        start = {k: v.clone() for k, v in model.state_dict().items()}
        # "adagradp": Adagrad with every accumulator preset to ADAGRAD_SUM0 -- from zero accumulators the first steps are
        # lr * sign(g), so an element whose gradient cancels to ~1e-8 is rounding noise in ANY implementation (AFM's
        # attention path); with a preset accumulator the step is lr * g / sqrt(s0 + g^2), smooth in g
        for opt_name in ("sgd", "adagrad", "adagradp"):
            model.load_state_dict(start)
            model.compile("adagrad" if opt_name == "adagradp" else opt_name, "binary_crossentropy", metrics=[])
            if opt_name == "adagradp":
                for grp in model.optim.param_groups:
                    for p in grp["params"]:
                        model.optim.state[p]["sum"].fill_(ADAGRAD_SUM0)
            losses = []
            for Xb, yb in zip(Xs, ys):  # the reference's own step, basemodel.py:242-262
                yp = model(torch.from_numpy(Xb)).squeeze()
                model.optim.zero_grad()
                ls = model.loss_func(yp, torch.from_numpy(yb), reduction="sum")
                total = ls + model.get_regularization_loss() + model.aux_loss
                total.backward()
                model.optim.step()
                losses.append(ls.item())
            out[opt_name + "3_loss"] = np.array(losses, np.float64)
            for k, v in model.state_dict().items():
                out[opt_name + "3/" + k] = v.detach().numpy().copy()
    if case_.get("lazy"):
        Xs, ys = zip(*[synth_inputs(spec, batch, rng) for _ in range(LAZY_STEPS)])
        out["lazy_X"], out["lazy_y"] = np.stack(Xs), np.stack(ys)
        start = {k: v.clone() for k, v in model.state_dict().items()}
        for tag, opt_name, l2 in (("sgd", "sgd", LAZY_L2), ("adagrad", "adagrad", LAZY_L2), ("adam", "adam", LAZY_L2),
                                  ("adam0", "adam", 0.0)):
            torch.manual_seed(case_["seed"])
            m = build_reference_model(ref, spec, l2=l2)
            m.load_state_dict(start)
            m.compile(opt_name, "binary_crossentropy", metrics=[])
            m.train()
            bce, tot = [], []
            for Xb, yb in zip(Xs, ys):  # the reference's own step, basemodel.py:242-262
                yp = m(torch.from_numpy(Xb)).squeeze()
                m.optim.zero_grad()
                ls = m.loss_func(yp, torch.from_numpy(yb), reduction="sum")
                total = ls + m.get_regularization_loss() + m.aux_loss
                total.backward()
                m.optim.step()
                bce.append(ls.item())
                tot.append(total.item())
            out["lazy_%s_bce" % tag] = np.array(bce, np.float64)
            out["lazy_%s_total" % tag] = np.array(tot, np.float64)
            for k, v in m.state_dict().items():
                out["lazy_%s/%s" % (tag, k)] = v.detach().numpy().copy()
            m.eval()
            with torch.no_grad():
                out["lazy_%s_pred" % tag] = m(torch.from_numpy(Xs[0])).numpy().reshape(-1, 1)
            # optimizer state of the first deep table (state_dict compatibility of the lazily updated state)
            p0 = m.embedding_dict[spec["dnn_columns"][0]["embedding_name"]].weight
            st = m.optim.state[p0]
            for key in ("sum", "exp_avg", "exp_avg_sq"):
                if key in st:
                    out["lazy_%s_state_%s" % (tag, key)] = st[key].detach().numpy().copy()
    if case_.get("fit"):
        Xf, yf = synth_inputs(spec, FIT_ROWS, rng)
        out["fit_X"], out["fit_y"] = Xf, yf
        names_ = [c["name"] for c in spec["dnn_columns"]]
        xin = {n: Xf[:, i] for i, n in enumerate(names_)}          # one column per feature (no VarLen here)
        start = {k: v.clone() for k, v in model.state_dict().items()}
        for tag, opt_name, l2, shuffle in FIT_RUNS:
            torch.manual_seed(case_["seed"])
            m = build_reference_model(ref, spec, l2=l2)
            m.load_state_dict(start)
            m.compile(opt_name, "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
            torch.manual_seed(FIT_SEED)
            hist = m.fit(xin, yf, batch_size=batch, epochs=FIT_EPOCHS, verbose=2, validation_split=FIT_SPLIT,
                         shuffle=shuffle)
            for k, v in hist.history.items():
                out["fit_%s_hist/%s" % (tag, k)] = np.asarray(v, np.float64)
            out["fit_%s_pred" % tag] = m.predict(xin, batch_size=50)
    return out


def main(names=None):
    ref = import_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for c in CASES:
        if names and c["name"] not in names:
            continue
        data = run_case(ref, c)
        path = os.path.join(GOLDEN_DIR, c["name"] + ".npz")
        np.savez_compressed(path, **data)
        print("%-24s B=%-3d logit[min,max]=[%+.3f,%+.3f] loss=%.4f  -> %s (%.0f KB)" % (
            c["name"], c["batch"], data["logit"].min(), data["logit"].max(), float(data["loss"]),
            os.path.relpath(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main(sys.argv[1:] or None)
