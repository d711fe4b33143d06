"""GPU parity of the MFMA tower (csrc/mlp.hip), the BCE head and the dense optimizer (csrc/head.hip).

Checkers: the numpy oracle's ``dnn_forward`` / ``dnn_backward`` / ``bce_sum`` (oracle/np_oracle.py, pinned to the
reference by the golden fixtures) evaluated in fp64, and a plain PyTorch fp32 restatement of the same ops on the
GPU (these are floating-point kernels).  Tolerances: 1e-5 x max|reference| on every output / gradient (fp32
re-association only: the MFMA f32 instructions are exact fmaf chains)."""
import copy
import pickle

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import build_model, golden_names, load_golden, max_abs
from np_oracle import bce_sum, dnn_backward, dnn_forward, sigmoid

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def _close(tag, got, ref, tol=TOL):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
    err = max_abs(got, ref)
    assert err <= tol * scale, "%s: max|d| = %.3e (scale %.3e)" % (tag, err, scale)


def _tower_modules(K, hidden, act="relu", seed=0):
    from deepctr_torch.layers import DNN
    torch.manual_seed(seed)
    dnn = DNN(K, hidden, activation=act, init_std=0.05, device=DEV)
    lin = torch.nn.Linear(hidden[-1], 1, bias=False).to(DEV)
    with torch.no_grad():
        for fc in dnn.linears:
            fc.bias.normal_(0, 0.05)
    return dnn, lin


SHAPES = [  # B, K, hidden, with projection, activation, extra input padding
    (37, 13, (8,), True, "relu", 0),
    (64, 429, (256, 128), True, "relu", 3),       # the DeepFM tower, input ld 432
    (4096, 429, (256, 128), True, "relu", 3),
    (100, 754, (64, 32, 16), True, "relu", 2),    # PNN-sized input (two K chunks), three layers
    (50, 40, (24, 20), False, "relu", 0),         # DCN: the tower output is the last hidden layer
    (33, 18, (10, 7), True, "linear", 2),         # identity activations, ragged widths
    (16, 5, (300,), True, "relu", 3),             # wide single layer: several tile passes per wave
    (70, 429, (1024, 512, 256), True, "relu", 3), # 1024-wide tower: the input is staged in 256-column chunks
    (40, 1100, (1152, 64), True, "relu", 0),      # the widest layer the LDS tiles hold, input in 64-column chunks
]


@pytest.mark.parametrize("B,K,hidden,proj,act,pad", SHAPES)
def test_tower_forward_backward_match_oracle_and_torch(B, K, hidden, proj, act, pad):
    from deepctr_torch._hip import mlp
    dnn, lin = _tower_modules(K, hidden, act)
    g = torch.Generator().manual_seed(1)
    xfull = torch.randn(B, K + pad, generator=g).to(DEV)     # the tower reads the first K columns
    xfull.requires_grad_(True)
    y = mlp.tower(dnn, lin if proj else None, xfull, K)
    gy = torch.randn(y.shape, generator=g)
    # a pre-activation within fp32 rounding of 0 may take the other relu branch than the fp64 oracle: such rows
    # get no upstream gradient, so the comparison never depends on a coin flip
    h64 = xfull.detach().cpu().numpy().astype(np.float64)[:, :K]
    risky = np.zeros(B, bool)
    for fc in dnn.linears:
        pre = h64 @ fc.weight.detach().cpu().numpy().astype(np.float64).T + fc.bias.detach().cpu().numpy().astype(np.float64)
        risky |= (np.abs(pre) < 1e-5).any(axis=1)
        h64 = np.maximum(pre, 0) if act == "relu" else pre
    gy[torch.from_numpy(risky)] = 0
    gy = gy.to(DEV)
    y.backward(gy)
    torch.cuda.synchronize()
    got = {"y": y.detach().cpu().numpy(), "gx": xfull.grad[:, :K].cpu().numpy()}
    for n_, p in list(dnn.named_parameters()) + [("out", lin.weight)]:
        got[n_] = p.grad.cpu().numpy() if p.grad is not None else None
    # ---- numpy oracle in fp64 ----
    P = {"dnn." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in dnn.named_parameters()}
    x64 = xfull.detach().cpu().numpy().astype(np.float64)[:, :K]
    if act == "relu":
        h, acts = dnn_forward(x64, P, "dnn.", len(hidden))
        w = lin.weight.detach().cpu().numpy().astype(np.float64)
        y_ref = h @ w.T if proj else h
        gy64 = gy.cpu().numpy().astype(np.float64)
        grads = {}
        gh = gy64 @ w if proj else gy64
        gx_ref = dnn_backward(gh, acts, P, "dnn.", len(hidden), grads)
        _close("y/oracle", got["y"], y_ref)
        _close("gx/oracle", got["gx"], gx_ref)
        for k, v in grads.items():
            _close(k + "/oracle", got[k[4:]], v)
        if proj:
            _close("w_out/oracle", got["out"], gy64.T @ h)
    # ---- plain PyTorch fp32 on the GPU ----
    dnn2, lin2 = copy.deepcopy(dnn), copy.deepcopy(lin)
    for p in list(dnn2.parameters()) + list(lin2.parameters()):
        p.grad = None
    x2 = xfull.detach()[:, :K].clone().requires_grad_(True)
    h2 = x2
    for fc in dnn2.linears:
        h2 = F.linear(h2, fc.weight, fc.bias)
        if act == "relu":
            h2 = torch.relu(h2)
    y2 = lin2(h2) if proj else h2
    y2.backward(gy)
    _close("y/torch", got["y"], y2.detach().cpu().numpy())
    _close("gx/torch", got["gx"], x2.grad.cpu().numpy())
    for (n_, p2) in dnn2.named_parameters():
        _close(n_ + "/torch", got[n_], p2.grad.cpu().numpy())
    if proj:
        _close("w_out/torch", got["out"], lin2.weight.grad.cpu().numpy())
    else:
        assert got["out"] is None
    # padding columns of the input receive no gradient garbage that autograd could see
    assert xfull.grad.shape == xfull.shape


def test_tower_is_bit_reproducible():
    from deepctr_torch._hip import mlp
    dnn, lin = _tower_modules(429, (256, 128))
    x = torch.randn(1024, 432, device=DEV)
    outs = []
    for _ in range(2):
        for p in list(dnn.parameters()) + [lin.weight]:
            p.grad = None
        xr = x.clone().requires_grad_(True)
        y = mlp.tower(dnn, lin, xr, 429)
        y.sum().backward()
        outs.append([y.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in dnn.parameters()] + [lin.weight.grad.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_tower_falls_back_to_modules_outside_its_envelope():
    """BatchNorm / Dice towers keep the PyTorch-ROCm module path (still on the GPU)."""
    from deepctr_torch._hip import mlp
    from deepctr_torch.layers import DNN
    dnn = DNN(12, (8, 4), use_bn=True, device=DEV)
    assert mlp.tower_layers(dnn, None) is None
    dnn = DNN(12, (8, 4), activation="prelu", device=DEV)
    assert mlp.tower_layers(dnn, None) is None
    y = mlp.tower(dnn, None, torch.randn(5, 12, device=DEV))
    assert y.shape == (5, 4)


@pytest.mark.parametrize("B", [1, 63, 4096, 20000])
@pytest.mark.parametrize("nparts", [1, 3, 4])
def test_bce_head_matches_oracle_and_torch(B, nparts):
    from deepctr_torch._hip import mlp
    g = torch.Generator().manual_seed(B + nparts)
    parts = [(torch.randn(B, 1, generator=g) * 2).to(DEV).requires_grad_(True) for _ in range(nparts)]
    if B > 8:
        with torch.no_grad():
            parts[0][0] = 40.0       # saturated sigmoid: exercises the -100 log clamp and the 1e-12 floor
            parts[0][1] = -40.0
    bias = torch.tensor([0.3], device=DEV, requires_grad=True)
    y = torch.randint(0, 2, (B,), generator=g).float().to(DEV)
    loss, y_pred = mlp.bce_head(parts, bias, y)
    (loss * 1.5).backward()
    # torch
    parts2 = [p.detach().clone().requires_grad_(True) for p in parts]
    bias2 = bias.detach().clone().requires_grad_(True)
    z = parts2[0]
    for p in parts2[1:]:
        z = z + p
    yp2 = torch.sigmoid(z + bias2).squeeze(1)
    loss2 = F.binary_cross_entropy(yp2, y, reduction="sum")
    (loss2 * 1.5).backward()
    _close("y_pred", y_pred.cpu().numpy(), yp2.detach().cpu().numpy(), 1e-6)
    assert abs(loss.item() - loss2.item()) <= 1e-5 * max(1.0, abs(loss2.item()))
    for a, b in zip(parts, parts2):
        _close("g_logit", a.grad.cpu().numpy(), b.grad.cpu().numpy(), 2e-6)
    assert abs(bias.grad.item() - bias2.grad.item()) <= 2e-5 * max(1.0, abs(bias2.grad.item()))
    # oracle (fp64)
    z64 = sum(p.detach().cpu().numpy().astype(np.float64) for p in parts).reshape(-1) + 0.3
    l64 = bce_sum(sigmoid(z64), y.cpu().numpy().astype(np.float64))
    if B <= 8:   # without the saturated rows fp32 and fp64 agree closely
        assert abs(loss.item() - float(l64)) <= 1e-5 * max(1.0, abs(float(l64)))


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
@pytest.mark.parametrize("n", [1, 7, 4096, 143875])
def test_dense_opt_matches_torch_optim(opt, n):
    import ctypes
    from deepctr_torch._hip import lib as L
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g).to(DEV)
    p_ref = torch.nn.Parameter(p0.clone())
    o = (torch.optim.SGD([p_ref], lr=0.01) if opt == "sgd" else torch.optim.Adagrad([p_ref], lr=0.01))
    p, st = p0.clone(), torch.zeros(n, device=DEV)
    for step in range(3):
        gr = torch.randn(n, generator=g).to(DEV)
        p_ref.grad = gr.clone()
        o.step()
        L.check(L.lib().dctr_dense_opt(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(gr.data_ptr()),
                                       ctypes.c_void_p(st.data_ptr()), n,
                                       L.UPD_SGD if opt == "sgd" else L.UPD_ADAGRAD, 0.01, 1e-10,
                                       L.stream_handle(DEV)))
    torch.cuda.synchronize()
    _close("param", p.cpu().numpy(), p_ref.detach().cpu().numpy(), 1e-6)


# ---- the fused train step through the drop-in model -----------------------------------------------------------
FUSABLE = [n for n in golden_names("deepfm") if "X_steps" in load_golden(n)["extra"]]


@pytest.mark.parametrize("name", FUSABLE)
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_fused_step_equals_autograd_step_and_reference(name, opt, monkeypatch):
    """DeepFM under compile('sgd'|'adagrad'), l2=0: the fused step (tower + head + slab optimizer) reproduces the
    reference's 3-step trajectory, and equals the autograd + torch.optim step it replaces."""
    g = load_golden(name)
    runs = {}
    for fused in ("1", "1h", "0"):      # fused step with the tower+head kernel, with separate tower / head, autograd
        monkeypatch.setenv("DCTR_FUSED_STEP", fused[0])
        monkeypatch.setenv("DCTR_FUSED_HEAD", "0" if fused == "1h" else "1")
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        losses = []
        for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
            loss, _, y_pred = m._train_step(torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV))
            losses.append(float(loss.item()))
        torch.cuda.synchronize()
        m.model_plan().check_ids()
        took_fused = bool(m._fused and m._fused.get("ok"))
        runs[fused] = (m, losses, took_fused)
    m1, l1, f1 = runs["1"]
    m0, l0, f0 = runs["0"]
    assert not f0
    if not f1:
        pytest.skip("%s is outside the fused step's envelope (no DNN / pooled fields)" % name)
    sd1, sd0, sdh = m1.state_dict(), m0.state_dict(), runs["1h"][0].state_dict()
    for k in sd0:
        _close("fused vs autograd: " + k, sd1[k].cpu().numpy(), sd0[k].cpu().numpy(), 2e-5)
        _close("fused (separate head) vs autograd: " + k, sdh[k].cpu().numpy(), sd0[k].cpu().numpy(), 2e-5)
    ref = g["extra"]
    key = opt + "3/"
    n_ref = 0
    for k in sd1:
        if key + k in ref:
            n_ref += 1
            assert max_abs(sd1[k].cpu().numpy(), ref[key + k]) <= 2e-5, "fused vs reference: " + k
    assert n_ref == len(sd1)
    np.testing.assert_allclose(l1, ref[opt + "3_loss"], rtol=2e-5)
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b))


def test_fused_step_keeps_state_dict_optimizer_state_and_pickle():
    name = "deepfm_criteo"
    g = load_golden(name)
    m = build_model(g["spec"], DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    keys, shapes = list(m.state_dict().keys()), [tuple(v.shape) for v in m.state_dict().values()]
    m.compile("adagrad", "binary_crossentropy", metrics=[])
    m.train()
    X, y = torch.from_numpy(g["extra"]["X_steps"][0]).to(DEV), torch.from_numpy(g["extra"]["y_steps"][0]).to(DEV)
    m._train_step(X, y)
    assert m._fused and m._fused["ok"], "deepfm_criteo must take the fused step"
    assert list(m.state_dict().keys()) == keys
    assert [tuple(v.shape) for v in m.state_dict().values()] == shapes
    # the torch optimizer's state IS the slab: its state_dict reflects the fused updates
    w = m.dnn.linears[0].weight
    assert float(m.optim.state[w]["sum"].abs().sum()) > 0
    osd = m.optim.state_dict()
    assert len(osd["state"]) == len(list(m.parameters()))
    # save / load round trip and whole-model pickle (reference tests/utils.py:162-170)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = build_model(g["spec"], DEV)
    m2.load_state_dict(sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k])
    m3 = pickle.loads(pickle.dumps(m))
    for (k, a), (_, b) in zip(m.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    # loading into the slab-seated model keeps the slab intact and the step keeps working
    m.load_state_dict(sd)
    assert m._fused["slab"].intact()
    m._train_step(X, y)
    torch.cuda.synchronize()


@pytest.mark.parametrize("double_buffer,spg", [(False, 1), (True, 1), (True, 2), (True, 3)])
def test_graphed_train_step_equals_eager(double_buffer, spg):
    """One hipGraph per static buffer set (copies of the next batch overlap the running graph): bit-identical to the
    eager fused step on the same sequence of batches (every kernel on the path is deterministic)."""
    from deepctr_torch._hip.graph import GraphedTrainStep
    g = load_golden("deepfm_criteo")
    gen = torch.Generator().manual_seed(5)
    Xs = [torch.from_numpy(g["extra"]["X_steps"][i % 3]).to(DEV)[torch.randperm(g["extra"]["X_steps"][0].shape[0], generator=gen).to(DEV)]
          for i in range(7)]
    ys = [torch.randint(0, 2, (Xs[0].shape[0],), generator=gen).float().to(DEV) for _ in range(7)]
    finals = []
    for mode in ("eager", "graph"):
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        losses = []
        m._train_step(Xs[0], ys[0])
        m._train_step(Xs[1], ys[1])
        step = m._train_step
        if mode == "graph":
            step = GraphedTrainStep(m, Xs[0], ys[0], steps_per_graph=spg, double_buffer=double_buffer).capture(Xs[2], ys[2])
        outs = []
        for i in range(2, 7):
            outs.append(step(Xs[i], ys[i]))
            if mode == "eager" or spg == 1:
                losses.append(float(outs[-1][0].item()))
        if mode == "graph":
            step.flush()             # 5 steps: an incomplete last group runs eagerly
        torch.cuda.synchronize()
        finals.append(({k: v.clone() for k, v in m.state_dict().items()}, losses))
    (a, la), (b, lb) = finals
    if spg == 1:
        assert la == lb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_graphed_step_block_equals_eager():
    """step_block(): a whole group of consecutive batches staged with two copies -- same parameters, bit for bit, as
    the eager fused step over the same batches (the weight gradients run on the fork stream in both)."""
    from deepctr_torch._hip.graph import GraphedTrainStep
    g = load_golden("deepfm_criteo")
    gen = torch.Generator().manual_seed(11)
    n = g["extra"]["X_steps"][0].shape[0]
    S = 3
    X_all = torch.cat([torch.from_numpy(g["extra"]["X_steps"][i % 3]) for i in range(2 + 2 * S)]).to(DEV)
    X_all = X_all[torch.randperm(X_all.shape[0], generator=gen).to(DEV)].contiguous()
    y_all = torch.randint(0, 2, (X_all.shape[0],), generator=gen).float().to(DEV)
    batch = lambda i: (X_all[i * n:(i + 1) * n], y_all[i * n:(i + 1) * n])
    finals = []
    for mode in ("eager", "graph"):
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        m._train_step(*batch(0))
        m._train_step(*batch(1))
        last = None
        if mode == "eager":
            for i in range(2, 2 + 2 * S):
                last = m._train_step(*batch(i))
        else:
            gs = GraphedTrainStep(m, *batch(0), steps_per_graph=S).capture(*batch(2))
            for grp in range(2):
                lo = 2 + grp * S
                last = gs.step_block(X_all[lo * n:(lo + S) * n], y_all[lo * n:(lo + S) * n])
        torch.cuda.synchronize()
        finals.append(({k: v.clone() for k, v in m.state_dict().items()}, float(last[0].item())))
    (a, la), (b, lb) = finals
    assert la == lb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_fit_replays_graphs_for_full_batches_and_matches_eager_fit(monkeypatch):
    """fit(): full-size batches replay the captured step, the ragged last batch runs eagerly; the trained parameters
    and the History are identical to a fit with graphs switched off (DCTR_FIT_GRAPH=0)."""
    g = load_golden("deepfm_criteo")
    names = [c["name"] for c in g["spec"]["dnn_columns"]]
    Xs = np.concatenate(list(g["extra"]["X_steps"]) + [g["X"]], axis=0)
    ys = np.concatenate(list(g["extra"]["y_steps"]) + [g["y"]], axis=0)
    n = (Xs.shape[0] // 48) * 48 + 17                     # 48-row batches + a ragged tail
    n = min(n, Xs.shape[0])
    x = {nm: Xs[:n, i] for i, nm in enumerate(names)}
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("DCTR_FIT_GRAPH", flag)
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile("adagrad", "binary_crossentropy", metrics=["binary_crossentropy"])
        hist = m.fit(x, ys[:n], batch_size=48, epochs=2, verbose=2, shuffle=False)
        used = m._fit_graph is not None and m._fit_graph.get("graph") is not None
        runs.append(({k: v.clone() for k, v in m.state_dict().items()}, dict(hist.history), used))
    (a, ha, ua), (b, hb, ub) = runs
    assert ua and not ub
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert ha["loss"] == hb["loss"] and ha["binary_crossentropy"] == hb["binary_crossentropy"]
