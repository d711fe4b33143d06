"""GPU: the fp32-MFMA CIN layer kernels (csrc/cin.hip) against the numpy oracle's statement of one CIN layer in fp64
(oracle/np_oracle.py cin_layer_forward / cin_layer_backward: the reference's einsum + 1x1 conv, interaction.py:216-229 -- the
functions np_oracle.cin_forward / cin_backward are made of, which tests/test_oracle_golden.py pins to the reference's xDeepFM
goldens), forward values at 1e-5 x scale and all four gradients at 2e-5 x scale."""
import numpy as np
import pytest
import torch

from np_oracle import cin_backward, cin_forward, cin_layer_backward, cin_layer_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n(t):
    return None if t is None else t.detach().double().cpu().numpy()


def _oracle_layer(H, X0, W, b, relu, R, same=False):
    """(A, gH, gX0, gW, gb) of  sum(A * R)  in fp64 by the oracle; ``same``: H is X0 (both gradients land on X0)."""
    Hn, Xn, Wn, bn = _n(H), _n(X0), _n(W), _n(b)
    A, cache = cin_layer_forward(Hn, Xn, Wn, bn, relu)
    gH, gX0, gW, gb = cin_layer_backward(_n(R), Hn, Xn, Wn, cache, relu)
    if same:
        gX0 = gX0 + gH
    return A, gH, gX0, gW, gb


def _check(name, got, want, tol):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got.detach().double().cpu().numpy() - want).max())
    assert err <= tol * scale, "%s: max|d|=%.3e (scale %.3g)" % (name, err, scale)


CASES = [  # B, h, M, D, O, relu, bias
    (5, 3, 3, 4, 8, True, True), (33, 7, 5, 16, 40, False, True), (64, 26, 26, 16, 128, True, True),
    (100, 64, 26, 16, 128, True, True), (17, 6, 4, 8, 200, True, False), (40, 2, 31, 5, 32, True, True),
    (257, 5, 7, 3, 33, False, False), (16, 26, 26, 16, 256, True, True),
    # 26 fields, wide outputs: the backward with rows flattened over (h, m) (k_cin_bwd_data_flat; 3, 13 + 2 and 1 tile periods,
    # a last tile past K, output chunks of 96 / 128 / 128 + 72)
    (100, 64, 26, 16, 128, True, True), (37, 3, 26, 8, 96, False, True), (50, 17, 26, 16, 200, True, False), (64, 1, 26, 4, 128, True, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_h%d_M%d_D%d_O%d" % c[:5])
def test_cin_layer_forward_backward(case):
    from deepctr_torch._hip.ops import CINLayerFunction
    B, h, M, D, O, relu, has_bias = case
    g = torch.Generator(device=DEV).manual_seed(B * 7 + O)
    H = (torch.randn(B, h, D, device=DEV, generator=g) * 0.5).requires_grad_(True)
    X0 = (torch.randn(B, M, D, device=DEV, generator=g) * 0.5).requires_grad_(True)
    W = (torch.randn(O, h * M, device=DEV, generator=g) * 0.1).requires_grad_(True)
    b = (torch.randn(O, device=DEV, generator=g) * 0.1).requires_grad_(True) if has_bias else None
    R = torch.randn(B, O, D, device=DEV, generator=g)
    A = CINLayerFunction.apply(H, X0, W, b, relu)
    (A * R).sum().backward()
    A2, gH, gX0, gW, gb = _oracle_layer(H, X0, W, b, relu, R)
    _check("A", A, A2, 1e-5)
    for name, a, r in (("gH", H.grad, gH), ("gX0", X0.grad, gX0), ("gW", W.grad, gW)) + ((("gb", b.grad, gb),) if has_bias else ()):
        _check(name, a, r, 2e-5)


SYM_CASES = [  # B, M, D, O, relu, bias     (H IS X0: the first layer of every CIN)
    (64, 26, 16, 128, True, True), (33, 5, 16, 40, False, True), (17, 31, 8, 200, True, False), (300, 26, 16, 256, True, True),
    (9, 1, 4, 8, True, True), (40, 2, 5, 32, False, False),
]


@pytest.mark.parametrize("case", SYM_CASES, ids=lambda c: "B%d_M%d_D%d_O%d" % c[:4])
def test_cin_first_layer_symmetric_products(case):
    """hidden state == field matrix (interaction.py:216-219 at i = 0): the kernels fold W[o, h, m] + W[o, m, h] and walk
    only the pairs h <= m forward and in the weight gradient; same bar as the general layer."""
    from deepctr_torch._hip.ops import CINLayerFunction
    B, M, D, O, relu, has_bias = case
    g = torch.Generator(device=DEV).manual_seed(B * 11 + O)
    X0 = (torch.randn(B, M, D, device=DEV, generator=g) * 0.5).requires_grad_(True)
    W = (torch.randn(O, M * M, device=DEV, generator=g) * 0.1).requires_grad_(True)
    b = (torch.randn(O, device=DEV, generator=g) * 0.1).requires_grad_(True) if has_bias else None
    R = torch.randn(B, O, D, device=DEV, generator=g)
    A = CINLayerFunction.apply(X0, X0, W, b, relu)
    (A * R).sum().backward()
    A2, _, gX0, gW, gb = _oracle_layer(X0, X0, W, b, relu, R, same=True)
    _check("A", A, A2, 1e-5)
    for name, a, r in (("gX0", X0.grad, gX0), ("gW", W.grad, gW)) + ((("gb", b.grad, gb),) if has_bias else ()):
        _check(name, a, r, 2e-5)


def test_cin_layer_on_strided_views():
    """X0 as a view of the gather's [B, ld] output, H as the first half of a previous layer's maps."""
    from deepctr_torch._hip.ops import CINLayerFunction
    B = 64
    out = torch.randn(B, 26 * 16 + 13 + 3, device=DEV)
    X0 = out[:, :416].reshape(B, 26, 16)
    prev = torch.randn(B, 128, 16, device=DEV)
    H = prev[:, :64]
    W = torch.randn(128, 64 * 26, device=DEV) * 0.05
    A = CINLayerFunction.apply(H, X0, W, None, True)
    _check("A", A, cin_layer_forward(_n(H), _n(X0), _n(W), None, True)[0], 1e-5)


@pytest.mark.parametrize("F,D,layers,split_half,B", [(26, 16, (128, 128), True, 96), (26, 16, (64, 32, 16), False, 40),
                                                     (7, 8, (10, 6), True, 33), (5, 4, (8,), True, 9), (31, 5, (12, 12), False, 17)])
def test_cin_module_matches_the_oracle_stack(F, D, layers, split_half, B):
    """The CIN module (layers/interaction.py: symmetric first layer, split_half hand-over, the pooling kernels) against
    np_oracle.cin_forward / cin_backward (interaction.py:207-248) in fp64: output and every gradient."""
    from deepctr_torch.layers import CIN
    torch.manual_seed(F * 7 + D)
    cin = CIN(F, layers, split_half=split_half, device=DEV)
    for p_ in cin.parameters():
        torch.nn.init.normal_(p_, 0, 0.1)
    X0 = (torch.randn(B, F, D, device=DEV) * 0.5).requires_grad_(True)
    R = torch.randn(B, sum(s // 2 if (split_half and i != len(layers) - 1) else s for i, s in enumerate(layers)), device=DEV)
    y = cin(X0)
    (y * R).sum().backward()
    P = {"cin." + k: _n(v) for k, v in cin.state_dict().items()}
    y2, cache = cin_forward(_n(X0), P, "cin.", layers, split_half)
    grads = {}
    gX0 = cin_backward(_n(R), _n(X0), cache, P, "cin.", layers, split_half, grads)
    _check("y", y, y2, 1e-5)
    _check("gX0", X0.grad, gX0, 2e-5)
    for k, p_ in cin.named_parameters():
        _check(k, p_.grad, grads["cin." + k].reshape(tuple(p_.shape)), 2e-5)


@pytest.mark.parametrize("F,D,layers,split_half,B,extra,relu", [
    (26, 16, (128, 128), True, 96, 13, True), (26, 16, (64, 32, 16), False, 40, 0, True), (7, 8, (10, 6), True, 33, 5, False),
    (5, 4, (8,), True, 9, 3, True), (31, 5, (12, 12), False, 17, 1, True), (26, 16, (128, 128), True, 1, 13, True)])
def test_cin_stack_with_projection_on_the_row_matrix(F, D, layers, split_half, B, extra, relu):
    """CIN + xDeepFM's bias-free 1-unit projection (xdeepfm.py:72, :97) as one autograd node on the gather's own
    [B, F*D + dense] row matrix (a view with a longer leading dimension, like the gather's padded output): the logit, the
    gradient in the row matrix's shape (zeros behind the fields), the layers' and the projection's weight gradients --
    against np_oracle.cin_forward / cin_backward and the projection written out in fp64."""
    from deepctr_torch.layers import CIN
    torch.manual_seed(F * 5 + D + B)
    cin = CIN(F, layers, activation="relu" if relu else "linear", split_half=split_half, device=DEV)
    for p_ in cin.parameters():
        torch.nn.init.normal_(p_, 0, 0.1)
    fm = sum(s // 2 if (split_half and i != len(layers) - 1) else s for i, s in enumerate(layers))
    w_head = (torch.randn(1, fm, device=DEV) * 0.3).requires_grad_(True)
    store = torch.randn(B, F * D + extra + 3, device=DEV) * 0.5
    x = store[:, :F * D + extra].detach().requires_grad_(True)      # (a non-contiguous leaf: stride F*D + extra + 3)
    assert x.stride(0) == F * D + extra + 3
    R = torch.randn(B, 1, device=DEV)
    y = cin._stack(x, F, D, w_head)
    assert y.shape == (B, 1)
    (y * R).sum().backward()
    P = {"cin." + k: _n(v) for k, v in cin.state_dict().items()}
    X0 = _n(x)[:, :F * D].reshape(B, F, D)
    act = "relu" if relu else "linear"
    out, cache = cin_forward(X0, P, "cin.", layers, split_half, activation=act)
    wn = _n(w_head)
    _check("logit", y, out @ wn.T, 1e-5)
    grads = {}
    g_feat = _n(R) @ wn
    gX0 = cin_backward(g_feat, X0, cache, P, "cin.", layers, split_half, grads, activation=act)
    want_gx = np.concatenate([gX0.reshape(B, F * D), np.zeros((B, extra))], axis=1)
    assert x.grad.shape == x.shape
    _check("gx", x.grad, want_gx, 2e-5)
    _check("g_w_head", w_head.grad, _n(R).T @ out, 2e-5)
    for k, p_ in cin.named_parameters():
        _check(k, p_.grad, grads["cin." + k].reshape(tuple(p_.shape)), 2e-5)


def test_xdeepfm_takes_the_stack_and_hooks_take_the_modules():
    """xDeepFM routes CIN + cin_linear through the one-node stack; a forward hook on either module sends the step
    through the modules again (so the hook fires) with the same logit."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import xDeepFM
    cols = [SparseFeat("c%d" % i, 20 + i, 4) for i in range(5)] + [DenseFeat("d", 2)]
    torch.manual_seed(2)
    m = xDeepFM(cols, cols, dnn_hidden_units=(16,), cin_layer_size=(8, 6), init_std=0.1, device=DEV)
    g = np.random.RandomState(0)
    X = torch.from_numpy(np.concatenate([g.randint(0, 20, (33, 5)), g.rand(33, 2)], axis=1).astype(np.float32)).to(DEV)
    m.eval()
    y0 = m(X)
    seen = []
    h = m.cin_linear.register_forward_hook(lambda mod, i, o: seen.append(o.shape))
    y1 = m(X)
    h.remove()
    assert seen == [(33, 1)]
    assert float((y0 - y1).detach().abs().max()) <= 1e-6


def test_cin_module_shapes_and_errors():
    from deepctr_torch.layers import CIN
    cin = CIN(5, (8, 6), device=DEV)
    assert [tuple(c.weight.shape) for c in cin.conv1ds] == [(8, 25, 1), (6, 20, 1)]
    y = cin(torch.randn(9, 5, 4, device=DEV))
    assert y.shape == (9, 4 + 6)
    with pytest.raises(ValueError):
        cin(torch.randn(9, 20, device=DEV))
    with pytest.raises(ValueError):
        CIN(5, (7, 6), split_half=True)
