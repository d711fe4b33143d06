"""GPU: the fp32-MFMA CIN layer kernels (csrc/cin.hip) against a plain PyTorch fp32/fp64 reference of the same op
(the reference's einsum + 1x1 conv, interaction.py:216-229), forward and all four gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(H, X0, W, b, relu):
    Z = (H[:, :, None, :] * X0[:, None, :, :]).reshape(H.shape[0], -1, H.shape[2])
    Y = torch.einsum("ok,bkd->bod", W, Z) + (b[None, :, None] if b is not None else 0)
    return torch.relu(Y) if relu else Y


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
    got = [A.detach(), H.grad, X0.grad, W.grad] + ([b.grad] if has_bias else [])
    H2, X2, W2 = (t.detach().double().requires_grad_(True) for t in (H, X0, W))
    b2 = b.detach().double().requires_grad_(True) if has_bias else None
    A2 = _ref(H2, X2, W2, b2, relu)
    (A2 * R.double()).sum().backward()
    want = [A2.detach(), H2.grad, X2.grad, W2.grad] + ([b2.grad] if has_bias else [])
    for name, a, r in zip(["A", "gH", "gX0", "gW", "gb"], got, want):
        scale = max(1.0, float(r.abs().max()))
        err = float((a.double() - r).abs().max())
        assert err <= 2e-5 * scale, "%s: max|d|=%.3e (scale %.3g)" % (name, err, scale)


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
    got = [A.detach(), X0.grad, W.grad] + ([b.grad] if has_bias else [])
    X2, W2 = (t.detach().double().requires_grad_(True) for t in (X0, W))
    b2 = b.detach().double().requires_grad_(True) if has_bias else None
    A2 = _ref(X2, X2, W2, b2, relu)
    (A2 * R.double()).sum().backward()
    want = [A2.detach(), X2.grad, W2.grad] + ([b2.grad] if has_bias else [])
    for name, a, r in zip(["A", "gX0", "gW", "gb"], got, want):
        scale = max(1.0, float(r.abs().max()))
        err = float((a.double() - r).abs().max())
        assert err <= 2e-5 * scale, "%s: max|d|=%.3e (scale %.3g)" % (name, err, scale)


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
    R = _ref(H.double(), X0.double(), W.double(), None, True)
    assert float((A.double() - R).abs().max()) <= 2e-5 * max(1.0, float(R.abs().max()))


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
