"""GPU: FiBiNET's pairs + first tower layer as one autograd node (csrc/bilinear_wide.hip, _hip/mlp.py
BilinearWideFunction) -- fibinet.py:82-99, interaction.py:140-156, core.py:123-133.

The checker is the numpy oracle in fp64 (np_oracle.bilinear_forward / bilinear_backward, pinned to the reference's FiBiNET
goldens by tests/test_oracle_golden.py) with the first layer ``relu(x W0^T + b0)`` written out beside it: every gradient at
2e-5 x scale.  The route that materialises the gradient slab (BilinearFunction + WideLinearFunction) must agree at the same
bar (its forward -- pair kernel + library GEMM -- within 2e-6 x scale of the fused forward's fixed-order sums), and two runs of
the fused node must give identical bits, forward and backward."""
import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n(t):
    return t.detach().double().cpu().numpy()


def _close(a, r, what, tol=2e-5):
    r = torch.from_numpy(np.ascontiguousarray(r)).to(a.device).reshape(a.shape) if isinstance(r, np.ndarray) else r
    scale = max(1.0, float(r.abs().max()))
    err = float((a.double() - r.double()).abs().max())
    assert err <= tol * scale, "%s: max|d|=%.3e (scale %.3g)" % (what, err, scale)


def _setup(B, F, D, H, nd, seed):
    from deepctr_torch.layers import BilinearInteraction, DNN
    torch.manual_seed(seed)
    layer = BilinearInteraction(F, D, "interaction", device=DEV)
    for p in layer.parameters():
        torch.nn.init.normal_(p, 0, 0.3)
    P = F * (F - 1) // 2
    dnn = DNN(2 * P * D + nd, (H, 8), device=DEV)
    for fc in dnn.linears:
        torch.nn.init.normal_(fc.weight, 0, 0.05)
        torch.nn.init.normal_(fc.bias, 0, 0.1)
    buf = torch.randn(B, F * D + nd + 3, device=DEV)          # like the gather's padded output
    E = buf[:, :F * D].reshape(B, F, D).detach().requires_grad_(True)
    V = (0.7 * torch.randn(B, F, D, device=DEV)).requires_grad_(True)
    dense = buf[:, F * D:F * D + nd].detach() if nd else None
    return layer, dnn, E, V, dense


def _run(layer, dnn, E, V, dense, R, lazy):
    from deepctr_torch._hip import mlp as _mlp
    for t in [E, V] + list(layer.parameters()) + list(dnn.parameters()):
        t.grad = None
    x = layer.fused_pair(E, V, dense, lazy=lazy)
    assert isinstance(x, _mlp.PendingPairs) == bool(lazy)
    W0, b0 = dnn.linears[0].weight, dnn.linears[0].bias
    if lazy:
        assert x.fits(W0, 1)
        h = _mlp.BilinearWideFunction.apply(x.meta, True, x.raw, x.senet, x.dense, W0, b0, *x.weights)
    else:
        h = _mlp.WideLinearFunction.apply(x, W0, b0, True)
    (h * R).sum().backward()
    return h.detach(), [E.grad.clone(), V.grad.clone(), W0.grad.clone(), b0.grad.clone()] + \
        [p.grad.clone() for p in layer.parameters()]


@pytest.mark.parametrize("B,F,H,nd", [(64, 26, 128, 13), (50, 6, 128, 3), (37, 26, 64, 0), (16, 5, 32, 2), (1, 4, 12, 1),
                                      (200, 9, 128, 5), (40, 9, 12, 0), (33, 12, 100, 7)])
def test_fused_node_against_the_oracle_and_the_slab_route(B, F, H, nd):
    D = 16
    layer, dnn, E, V, dense = _setup(B, F, D, H, nd, seed=B + F)
    R = torch.randn(B, H, device=DEV)
    h1, g1 = _run(layer, dnn, E, V, dense, R, lazy=True)
    h0, g0 = _run(layer, dnn, E, V, dense, R, lazy=False)
    _close(h1, h0, "h (fused forward against bilinear kernel + library GEMM)", tol=2e-6)
    # fp64 oracle
    P = F * (F - 1) // 2
    Pn = {"bl.bilinear.%d.weight" % k: _n(p) for k, p in enumerate(layer.parameters())}
    pv, pe = O.bilinear_forward(_n(V), Pn, "bl.", "interaction"), O.bilinear_forward(_n(E), Pn, "bl.", "interaction")
    parts = [pv.reshape(B, -1), pe.reshape(B, -1)] + ([_n(dense)] if nd else [])
    x = np.concatenate(parts, 1)
    W0, b0 = _n(dnn.linears[0].weight), _n(dnn.linears[0].bias)
    pre = x @ W0.T + b0
    _close(h1, np.maximum(pre, 0), "h", tol=1e-5)
    gh = _n(R) * (pre > 0)
    gx = gh @ W0
    grads = {}
    gVn = O.bilinear_backward(gx[:, :P * D].reshape(B, P, D), _n(V), Pn, "bl.", "interaction", grads)
    gEn = O.bilinear_backward(gx[:, P * D:2 * P * D].reshape(B, P, D), _n(E), Pn, "bl.", "interaction", grads)
    ref = [gEn, gVn, gh.T @ x, gh.sum(0)] + [grads["bl.bilinear.%d.weight" % k] for k in range(P)]
    names = ["gE", "gV", "gW0", "gb0"] + ["gW%d" % k for k in range(P)]
    for a, b, r, nm in zip(g1, g0, ref, names):
        _close(a, r, nm + " (fused, oracle)")
        _close(b, r, nm + " (slab route, oracle)")


def test_fused_backward_is_bit_reproducible_and_leaves_no_slab():
    B, F, D, H, nd = 300, 26, 16, 128, 13
    layer, dnn, E, V, dense = _setup(B, F, D, H, nd, seed=11)
    R = torch.randn(B, H, device=DEV)
    ha, ga = _run(layer, dnn, E, V, dense, R, lazy=True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    hb, gb = _run(layer, dnn, E, V, dense, R, lazy=True)
    assert torch.equal(ha, hb)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    # the backward holds the saved DNN input (one slab) and the per-tile weight partials, never a second [B, 2PD] slab
    slab = B * 2 * (F * (F - 1) // 2) * D * 4
    tiles = (B + 15) // 16
    assert torch.cuda.max_memory_allocated() - base < 1.5 * slab + tiles * 325 * 256 * 4 + (8 << 20)


def test_fibinet_takes_the_fused_node(monkeypatch):
    """The model's own step goes through dctr_bilinear_wide_bwd (and through the slab route with DCTR_BILINEAR_WIDE=0),
    same loss, gradients within the bar of two summation orders."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import FiBiNET
    from deepctr_torch._hip import lib as L
    # (17 fields of 16: a DNN input of 4 355 columns -- past the 4 096 the tower kernels take themselves)
    cols = [SparseFeat("s%d" % i, 50 + i, embedding_dim=16) for i in range(17)] + [DenseFeat("d%d" % i, 1) for i in range(3)]
    g = torch.Generator().manual_seed(5)
    X = torch.cat([torch.stack([torch.randint(0, 50 + i, (96,), generator=g) for i in range(17)], 1).float(),
                   torch.rand(96, 3, generator=g)], 1).to(DEV)
    y = (torch.rand(96, generator=g) < 0.4).float().to(DEV)
    calls = []
    lib = L.lib()
    real = lib.dctr_bilinear_wide_bwd

    def spy(*a):
        calls.append(1)
        return real(*a)

    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DCTR_BILINEAR_WIDE", mode)
        model = FiBiNET(cols, cols, dnn_hidden_units=(64, 16), init_std=0.1, seed=7, device=DEV)
        monkeypatch.setattr(lib, "dctr_bilinear_wide_bwd", spy)
        n0 = len(calls)
        model.train()
        out = model(X).reshape(-1)
        loss = torch.nn.functional.binary_cross_entropy(out, y.reshape(-1).float(), reduction="sum")
        loss.backward()
        assert (len(calls) > n0) == (mode == "1")
        res[mode] = (float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert abs(res["1"][0] - res["0"][0]) <= 2e-6 * abs(res["0"][0])      # (two summation orders of the first layer)
    assert res["1"][1].keys() == res["0"][1].keys()
    for n, a in res["1"][1].items():
        _close(a, res["0"][1][n], n, tol=2e-5)
