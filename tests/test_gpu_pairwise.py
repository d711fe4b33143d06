"""GPU: SENET / Bilinear / InnerProduct / CrossNet kernels (csrc/pairwise.hip, cross.hip) against the numpy oracle in fp64
(oracle/np_oracle.py senet_* / bilinear_* / inner_product_* / crossnet_*: what the reference's layers compute,
interaction.py:93-101,140-156,438-453,557-577 -- pinned to the reference's FiBiNET / PNN / DCN goldens by
tests/test_oracle_golden.py): values at 1e-5 x scale, every gradient at 2e-5 x scale.  (BiInteraction / AFM / Interacting /
CrossNetMix further down keep a torch fp64 expression beside their model-level goldens.)"""
import itertools

import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n(t):
    return t.detach().double().cpu().numpy()


def _close(a, r, what, tol=2e-5):
    if r is None:                      # torch leaves the gradient of an unused parameter undefined; we return zeros
        assert a is None or float(a.abs().max()) == 0.0, what
        return
    if isinstance(r, np.ndarray):      # an oracle value
        r = torch.from_numpy(np.ascontiguousarray(r)).to(a.device).reshape(a.shape)
    scale = max(1.0, float(r.abs().max())) if r.numel() else 1.0
    err = float((a.double() - r).abs().max()) if r.numel() else 0.0
    assert err <= tol * scale, "%s: max|d|=%.3e (scale %.3g)" % (what, err, scale)


@pytest.mark.parametrize("B,F,D,ratio", [(5, 3, 4, 1), (37, 26, 16, 3), (64, 10, 16, 3), (9, 5, 8, 2), (1, 2, 3, 5)])
def test_senet(B, F, D, ratio):
    from deepctr_torch.layers import SENETLayer
    torch.manual_seed(B + F)
    layer = SENETLayer(F, ratio, device=DEV)
    for p in layer.parameters():
        torch.nn.init.normal_(p, 0, 0.5)
    E = torch.randn(B, F, D, device=DEV, requires_grad=True)
    R = torch.randn(B, F, D, device=DEV)
    V = layer(E)
    (V * R).sum().backward()
    W1, W2 = _n(layer.excitation[0].weight), _n(layer.excitation[2].weight)
    V2, cache = O.senet_forward(_n(E), W1, W2)
    gE, gW1, gW2 = O.senet_backward(_n(R), _n(E), cache, W1, W2)
    _close(V.detach(), V2, "V", tol=1e-5)
    _close(E.grad, gE, "gE")
    _close(layer.excitation[0].weight.grad, gW1, "gW1")
    _close(layer.excitation[2].weight.grad, gW2, "gW2")


def _bilinear_ref(X, Ws, btype):
    F = X.shape[1]
    outs = []
    for k, (i, j) in enumerate(itertools.combinations(range(F), 2)):
        W = Ws[0] if btype == "all" else (Ws[i] if btype == "each" else Ws[k])
        outs.append((X[:, i] @ W.t()) * X[:, j])
    return torch.stack(outs, 1)


@pytest.mark.parametrize("btype", ["interaction", "each", "all"])
@pytest.mark.parametrize("B,F,D", [(7, 3, 4), (33, 5, 8), (48, 10, 16), (100, 26, 16), (16, 4, 5), (257, 7, 16)])
def test_bilinear_single_input(B, F, D, btype):
    from deepctr_torch.layers import BilinearInteraction
    torch.manual_seed(F * 31 + D)
    layer = BilinearInteraction(F, D, btype, device=DEV)
    for p in layer.parameters():
        torch.nn.init.normal_(p, 0, 0.3)
    X = torch.randn(B, F, D, device=DEV, requires_grad=True)
    P = F * (F - 1) // 2
    R = torch.randn(B, P, D, device=DEV)
    out = layer(X)
    assert out.shape == (B, P, D)
    (out * R).sum().backward()
    Pn = {"bl." + k: _n(v) for k, v in layer.state_dict().items()}
    ref = O.bilinear_forward(_n(X), Pn, "bl.", btype)
    grads = {}
    gX = O.bilinear_backward(_n(R), _n(X), Pn, "bl.", btype, grads)
    _close(out.detach(), ref, "out", tol=1e-5)
    _close(X.grad, gX, "gX")
    for k, p in layer.named_parameters():
        _close(p.grad, grads.get("bl." + k), "g" + k)      # ('each': the last field's matrix is never a left factor)


@pytest.mark.parametrize("btype", ["interaction", "each", "all"])
def test_bilinear_fused_pair_is_the_fibinet_dnn_input(btype):
    from deepctr_torch.layers import BilinearInteraction
    B, F, D, nd = 50, 6, 16, 3
    torch.manual_seed(3)
    layer = BilinearInteraction(F, D, btype, device=DEV)
    for p in layer.parameters():
        torch.nn.init.normal_(p, 0, 0.3)
    buf = torch.randn(B, F * D + nd + 1, device=DEV)          # like the gather's padded output
    E = buf[:, :F * D].reshape(B, F, D).detach().requires_grad_(True)
    V = torch.randn(B, F, D, device=DEV, requires_grad=True)
    dense = buf[:, F * D:F * D + nd].detach().requires_grad_(True)
    out = layer.fused_pair(E, V, dense)
    P = F * (F - 1) // 2
    assert out.shape == (B, 2 * P * D + nd)
    R = torch.randn_like(out)
    (out * R).sum().backward()
    params = list(layer.parameters())
    E2, V2, d2 = (t.detach().double().requires_grad_(True) for t in (E, V, dense))
    Ws = [p.detach().double().requires_grad_(True) for p in params]
    ref = torch.cat([_bilinear_ref(V2, Ws, btype).flatten(1), _bilinear_ref(E2, Ws, btype).flatten(1), d2], 1)
    (ref * R.double()).sum().backward()
    _close(out.detach(), ref.detach(), "out")
    _close(E.grad, E2.grad, "gE")
    _close(V.grad, V2.grad, "gV")
    _close(dense.grad, d2.grad, "gdense")
    for k, (p, w) in enumerate(zip(params, Ws)):
        _close(p.grad, w.grad, "gW%d" % k)
    # bit-reproducible
    for p in params:
        p.grad = None
    E.grad = V.grad = None
    out2 = layer.fused_pair(E, V, dense)
    (out2 * R).sum().backward()
    assert torch.equal(out2, out)
    E3g = E.grad.clone()
    E.grad = None
    for p in params:
        p.grad = None
    (layer.fused_pair(E, V, dense) * R).sum().backward()
    assert torch.equal(E.grad, E3g)


@pytest.mark.parametrize("reduce_sum", [True, False])
@pytest.mark.parametrize("B,F,D", [(5, 2, 4), (33, 8, 8), (100, 26, 16), (17, 5, 3)])
def test_inner_product(B, F, D, reduce_sum):
    from deepctr_torch.layers import InnerProductLayer
    torch.manual_seed(B)
    E = torch.randn(B, F, D, device=DEV, requires_grad=True)
    layer = InnerProductLayer(reduce_sum=reduce_sum, device=DEV)
    out = layer([E[:, f:f + 1] for f in range(F)])
    P = F * (F - 1) // 2
    assert out.shape == (B, P, 1 if reduce_sum else D)
    R = torch.randn_like(out)
    (out * R).sum().backward()
    if reduce_sum:          # the oracle's statement (interaction.py:557-577, reduce_sum=True: PNN's use)
        ref, pairs = O.inner_product_forward(_n(E))
        _close(out.detach(), ref[:, :, None], "out", tol=1e-5)
        _close(E.grad, O.inner_product_backward(_n(E), pairs, _n(R)[:, :, 0]), "gE")
        return
    E2 = E.detach().double().requires_grad_(True)
    row, col = zip(*itertools.combinations(range(F), 2))
    ref = E2[:, list(row)] * E2[:, list(col)]
    (ref * R.double()).sum().backward()
    _close(out.detach(), ref.detach(), "out")
    _close(E.grad, E2.grad, "gE")


@pytest.mark.parametrize("param", ["vector", "matrix"])
@pytest.mark.parametrize("B,W,L", [(3, 5, 1), (48, 69, 3), (100, 429, 2), (1000, 429, 2), (7, 845, 4)])
def test_crossnet(B, W, L, param):
    from deepctr_torch.layers import CrossNet
    torch.manual_seed(W)
    layer = CrossNet(W, L, param, device=DEV)
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
        layer.kernels.mul_(0.5)
    X = (torch.randn(B, W, device=DEV) * 0.5).requires_grad_(True)
    R = torch.randn(B, W, device=DEV)
    Y = layer(X)
    if W <= 512:        # both parameterisations run on this repo's kernels there (not on hipBLASLt)
        assert type(Y.grad_fn).__name__ in ("CrossNetVecFunctionBackward", "CrossNetMatFunctionBackward",
                                            "SliceBackward0", "AliasBackward0"), type(Y.grad_fn).__name__
        if param == "matrix":
            node = Y.grad_fn
            while type(node).__name__ != "CrossNetMatFunctionBackward":
                assert node.next_functions, "the matrix form did not go through dctr_crossnet_mat_fwd"
                node = node.next_functions[0][0]
    (Y * R).sum().backward()
    K, Bs = _n(layer.kernels), _n(layer.bias)
    ref, xs = O.crossnet_forward(_n(X), K, Bs, param)
    gX, gK, gB = O.crossnet_backward(_n(R), xs, K, Bs, param)
    _close(Y.detach(), ref, "Y", tol=1e-5)
    _close(X.grad, gX, "gX")
    _close(layer.kernels.grad, gK, "gK", tol=5e-5)
    _close(layer.bias.grad, gB, "gb")


@pytest.mark.parametrize("B,F,D", [(7, 3, 4), (64, 26, 16), (33, 5, 7)])
def test_bi_interaction_pooling_matches_torch(B, F, D):
    """BiInteractionPooling (interaction.py:54-61) standalone: forward and gradient against the reference's formula."""
    from deepctr_torch.layers import BiInteractionPooling
    g = torch.Generator(device=DEV).manual_seed(B + F)
    E = torch.randn(B, F, D, device=DEV, generator=g).requires_grad_(True)
    R = torch.randn(B, 1, D, device=DEV, generator=g)
    y = BiInteractionPooling()(E)
    assert y.shape == (B, 1, D)
    (y * R).sum().backward()
    E2 = E.detach().double().requires_grad_(True)
    y2 = 0.5 * (torch.pow(E2.sum(dim=1, keepdim=True), 2) - (E2 * E2).sum(dim=1, keepdim=True))
    (y2 * R.double()).sum().backward()
    assert float((y.double() - y2).abs().max()) <= 1e-5 * max(1.0, float(y2.abs().max()))
    assert float((E.grad.double() - E2.grad).abs().max()) <= 1e-5 * max(1.0, float(E2.grad.abs().max()))
    with pytest.raises(ValueError):
        BiInteractionPooling()(torch.randn(4, 8, device=DEV))


@pytest.mark.parametrize("B,F,D,A", [(9, 3, 4, 4), (64, 26, 16, 8), (33, 7, 5, 3), (1100, 10, 8, 8), (5, 2, 64, 32)])
def test_afm_layer_matches_torch(B, F, D, A):
    """AFMLayer (interaction.py:299-325): forward, input gradient and the four parameter gradients against the
    reference's own sequence of torch ops evaluated in fp64."""
    from deepctr_torch.layers import AFMLayer
    torch.manual_seed(B + F + D)
    layer = AFMLayer(D, attention_factor=A, device=DEV)
    with torch.no_grad():
        layer.attention_b.normal_(0, 0.3)
    g = torch.Generator(device=DEV).manual_seed(B)
    E = (torch.randn(B, F, D, device=DEV, generator=g) * 0.7).requires_grad_(True)
    R = torch.randn(B, 1, device=DEV, generator=g)
    y = layer([E[:, f:f + 1] for f in range(F)])            # the reference's calling convention: a list of [B, 1, D]
    assert y.shape == (B, 1)
    (y * R).sum().backward()
    got = [y.detach(), E.grad] + [p.grad for p in (layer.attention_W, layer.attention_b, layer.projection_h, layer.projection_p)]
    E2 = E.detach().double().requires_grad_(True)
    W, b, h, p = (t.detach().double().requires_grad_(True) for t in
                  (layer.attention_W, layer.attention_b, layer.projection_h, layer.projection_p))
    idx = torch.triu_indices(F, F, 1, device=DEV)
    bi = E2[:, idx[0]] * E2[:, idx[1]]
    att = torch.relu(torch.tensordot(bi, W, dims=([-1], [0])) + b)
    score = torch.softmax(torch.tensordot(att, h, dims=([-1], [0])), dim=1)
    y2 = torch.tensordot(torch.sum(score * bi, dim=1), p, dims=([-1], [0]))
    (y2 * R.double()).sum().backward()
    want = [y2.detach(), E2.grad, W.grad, b.grad, h.grad, p.grad]
    for name, a, r in zip(["y", "gE", "gW", "gb", "gh", "gp"], got, want):
        scale = max(1.0, float(r.abs().max()))
        err = float((a.double() - r).abs().max())
        assert err <= 2e-5 * scale, "%s: max|d|=%.3e (scale %.3g)" % (name, err, scale)


def test_afm_rejects_dense_on_the_deep_side():
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import AFM
    cols = [SparseFeat("a", 5, 4), SparseFeat("b", 6, 4), DenseFeat("d", 1)]
    with pytest.raises(ValueError, match="DenseFeat is not supported"):
        AFM(cols, cols, device=DEV)


@pytest.mark.parametrize("B,F,D,H,res,scaling", [(9, 3, 4, 2, True, False), (64, 26, 16, 2, True, False),
                                                 (33, 7, 8, 4, False, True), (1100, 10, 8, 1, True, True),
                                                 (5, 64, 32, 8, True, False), (6, 5, 40, 2, True, False)])
def test_interacting_layer_matches_torch(B, F, D, H, res, scaling):
    """InteractingLayer (interaction.py:366-394): forward, input gradient and the weight gradients against the
    reference's own sequence of torch ops in fp64.  The last case (D = 40) is outside the kernel: PyTorch-ROCm path."""
    from deepctr_torch.layers import InteractingLayer
    torch.manual_seed(B + F + D)
    layer = InteractingLayer(D, head_num=H, use_res=res, scaling=scaling, device=DEV)
    with torch.no_grad():
        for p in layer.parameters():
            p.normal_(0, 0.3)
    g = torch.Generator(device=DEV).manual_seed(B)
    E = (torch.randn(B, F, D, device=DEV, generator=g) * 0.7).requires_grad_(True)
    R = torch.randn(B, F, D, device=DEV, generator=g)
    y = layer(E)
    assert y.shape == (B, F, D)
    (y * R).sum().backward()
    names = ["W_Query", "W_key", "W_Value"] + (["W_Res"] if res else [])
    got = [y.detach(), E.grad] + [getattr(layer, n).grad for n in names]
    E2 = E.detach().double().requires_grad_(True)
    Ws = {n: getattr(layer, n).detach().double().requires_grad_(True) for n in names}
    A = D // H
    q, k, v = (torch.tensordot(E2, Ws[n], dims=([-1], [0])) for n in names[:3])
    q, k, v = (torch.stack(torch.split(t, A, dim=2)) for t in (q, k, v))
    inner = torch.einsum("bnik,bnjk->bnij", q, k)
    if scaling:
        inner = inner / A ** 0.5
    result = torch.matmul(torch.softmax(inner, dim=-1), v)
    result = torch.squeeze(torch.cat(torch.split(result, 1), dim=-1), dim=0)
    if res:
        result = result + torch.tensordot(E2, Ws["W_Res"], dims=([-1], [0]))
    y2 = torch.relu(result)
    (y2 * R.double()).sum().backward()
    want = [y2.detach(), E2.grad] + [Ws[n].grad for n in names]
    for name, a, r in zip(["y", "gE"] + names, got, want):
        scale = max(1.0, float(r.abs().max()))
        err = float((a.double() - r).abs().max())
        assert err <= 2e-5 * scale, "%s: max|d|=%.3e (scale %.3g)" % (name, err, scale)


@pytest.mark.parametrize("B,W,L,E,R", [(3, 5, 1, 2, 3), (33, 40, 3, 3, 5), (48, 69, 2, 3, 8), (48, 69, 2, 4, 32), (100, 429, 2, 4, 32),
                                       (1000, 429, 2, 4, 32), (64, 300, 4, 8, 16)])
def test_crossnet_mix(B, W, L, E, R):
    """CrossNetMix (interaction.py:499-534) on the kernels (dctr_crossnet_mix_*) against the reference's formulation in
    fp64: output and every gradient (x, U, V, C, the shared gating weights, bias)."""
    from deepctr_torch.layers import CrossNetMix
    torch.manual_seed(W + R)
    layer = CrossNetMix(W, low_rank=R, num_experts=E, layer_num=L, device=DEV)
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
        for gt in layer.gating:
            gt.weight.normal_(0, 0.3)
    X = (torch.randn(B, W, device=DEV) * 0.5).requires_grad_(True)
    Rm = torch.randn(B, W, device=DEV)
    Y = layer(X)
    node = Y.grad_fn
    while type(node).__name__ != "CrossNetMixFunctionBackward":
        assert node.next_functions, "CrossNetMix did not go through dctr_crossnet_mix_fwd"
        node = node.next_functions[0][0]
    (Y * Rm).sum().backward()
    # the reference's formulation (per-expert loop), fp64
    X2 = X.detach().double().requires_grad_(True)
    U = layer.U_list.detach().double().requires_grad_(True)
    V = layer.V_list.detach().double().requires_grad_(True)
    C = layer.C_list.detach().double().requires_grad_(True)
    Gs = [gt.weight.detach().double().requires_grad_(True) for gt in layer.gating]
    Bs = layer.bias.detach().double().requires_grad_(True)
    x_0 = X2.unsqueeze(2)
    x_l = x_0
    for i in range(L):
        outs, gates = [], []
        for e in range(E):
            gates.append(x_l.squeeze(2) @ Gs[e].t())
            v_x = torch.tanh(torch.matmul(V[i][e].t(), x_l))
            v_x = torch.tanh(torch.matmul(C[i][e], v_x))
            uv_x = torch.matmul(U[i][e], v_x)
            outs.append((x_0 * (uv_x + Bs[i])).squeeze(2))
        outs = torch.stack(outs, 2)
        gate = torch.stack(gates, 1)
        moe = torch.matmul(outs, gate.softmax(1))
        x_l = moe + x_l
    ref = x_l.squeeze(2)
    (ref * Rm.double()).sum().backward()
    _close(Y.detach(), ref.detach(), "Y", tol=2e-5)
    _close(X.grad, X2.grad, "gX", tol=5e-5)
    _close(layer.U_list.grad, U.grad, "gU", tol=5e-5)
    _close(layer.V_list.grad, V.grad, "gV", tol=5e-5)
    _close(layer.C_list.grad, C.grad, "gC", tol=5e-5)
    _close(layer.bias.grad, Bs.grad, "gb", tol=5e-5)
    for e in range(E):
        _close(layer.gating[e].weight.grad, Gs[e].grad, "gG%d" % e, tol=5e-5)
