"""CPU: the Python layer above the tower / head / dense-optimizer entry points (autograd Functions, DenseSlab,
sink route, argument marshalling), run against a numpy stand-in for the library (tests/mock_lib.py) and compared
with plain torch autograd.  The kernels themselves are covered by tests/test_gpu_mlp.py on the GPU."""
import copy

import pytest
import torch
import torch.nn.functional as F



def _modules(K, hidden, act="relu"):
    from deepctr_torch.layers import DNN
    torch.manual_seed(0)
    dnn = DNN(K, hidden, activation=act, init_std=0.1)
    lin = torch.nn.Linear(hidden[-1], 1, bias=False)
    return dnn, lin


def _torch_ref(dnn, lin, x, K, gy, act="relu"):
    dnn2, lin2 = copy.deepcopy(dnn), (copy.deepcopy(lin) if lin is not None else None)
    x2 = x.detach()[:, :K].clone().requires_grad_(True)
    h = x2
    for fc in dnn2.linears:
        h = F.linear(h, fc.weight, fc.bias)
        if act == "relu":
            h = torch.relu(h)
    y = lin2(h) if lin2 is not None else h
    y.backward(gy)
    return y.detach(), x2.grad, dnn2, lin2


@pytest.mark.parametrize("B,K,hidden,proj,act,pad", [(9, 13, (8, 5), True, "relu", 3), (7, 12, (6,), False, "relu", 0),
                                                    (5, 10, (4, 3), True, "linear", 2)])
def test_tower_function_autograd_route(mock, B, K, hidden, proj, act, pad):
    from deepctr_torch._hip import mlp
    dnn, lin = _modules(K, hidden, act)
    x = torch.randn(B, K + pad, requires_grad=True)
    y = mlp.tower(dnn, lin if proj else None, x, K)
    gy = torch.randn(y.shape)
    y.backward(gy)
    y2, gx2, dnn2, lin2 = _torch_ref(dnn, lin if proj else None, x, K, gy, act)
    assert torch.allclose(y, y2, atol=1e-5)
    assert x.grad.shape == x.shape and torch.allclose(x.grad[:, :K], gx2, atol=1e-5)
    for p, q in zip(dnn.parameters(), dnn2.parameters()):
        assert p.grad.shape == p.shape and torch.allclose(p.grad, q.grad, atol=1e-5)
    if proj:
        assert torch.allclose(lin.weight.grad, lin2.weight.grad, atol=1e-5)
    assert mock.calls == ["mlp_fwd", "mlp_bwd"]
    with torch.no_grad():          # inference: nothing is saved, no backward
        y3 = mlp.tower(dnn, lin if proj else None, x.detach(), K)
    assert torch.allclose(y3, y2, atol=1e-5)


def test_tower_sink_route_and_slab_optimizer(mock):
    """Parameters re-seated in the slab (row-padded weights), gradients written straight into the gradient slab,
    one dense_opt call == torch.optim on the same gradients."""
    from deepctr_torch._hip import dense, mlp
    K, hidden, B = 13, (8, 5), 11
    dnn, lin = _modules(K, hidden)
    bias = torch.nn.Parameter(torch.zeros(1))
    ref_dnn, ref_lin, ref_bias = copy.deepcopy(dnn), copy.deepcopy(lin), torch.nn.Parameter(torch.zeros(1))
    params = list(dnn.parameters()) + [lin.weight, bias]
    opt = torch.optim.Adagrad(params, lr=0.05)
    slab = dense.DenseSlab(params, pad_rows=[fc.weight for fc in dnn.linears])
    assert dnn.linears[0].weight.stride() == (16, 1) and dnn.linears[0].weight.shape == (8, 13)
    slab.adopt_adagrad_state(opt)
    slab.attach_grads()
    ref_opt = torch.optim.Adagrad(list(ref_dnn.parameters()) + [ref_lin.weight, ref_bias], lr=0.05)
    for step in range(3):
        x = torch.randn(B, 16)
        yv = torch.randint(0, 2, (B,)).float()
        extra = torch.randn(B, 1, requires_grad=True)
        logit = mlp.tower(dnn, lin, x.clone().requires_grad_(True), K, sink=slab)
        loss, y_pred = mlp.bce_head([extra, logit], bias, yv, unit=True, g_bias_sink=slab.grad_of(bias))
        loss.backward()
        slab.step("adagrad", 0.05, 1e-10)
        # reference
        h = x[:, :K]
        for fc in ref_dnn.linears:
            h = torch.relu(fc(h))
        yp = torch.sigmoid(extra.detach() + ref_lin(h) + ref_bias).squeeze(1)
        l2 = F.binary_cross_entropy(yp, yv, reduction="sum")
        ref_opt.zero_grad()
        l2.backward()
        ref_opt.step()
        assert abs(loss.item() - l2.item()) < 1e-4
        assert torch.allclose(extra.grad.squeeze(1), (yp - yv).detach(), atol=1e-5)
    for p, q in zip(list(dnn.parameters()) + [lin.weight, bias], list(ref_dnn.parameters()) + [ref_lin.weight, ref_bias]):
        assert torch.allclose(p, q, atol=1e-5), (p.shape, (p - q).abs().max())
    # the padding of the slab never moved and the optimizer's state lives in the slab
    W = dnn.linears[0].weight
    off, rows, cols, ld = slab._lay[id(W)]
    assert float(slab.flat[off:off + rows * ld].view(rows, ld)[:, cols:].abs().sum()) == 0.0
    assert opt.state[W]["sum"].data_ptr() == slab.state.data_ptr() + off * 4
    assert slab.intact()
    assert set(mock.calls) == {"mlp_fwd", "mlp_bwd", "bce_head", "dense_opt"}


def test_bce_head_autograd_route(mock):
    from deepctr_torch._hip import mlp
    B = 17
    parts = [torch.randn(B, 1, requires_grad=True) for _ in range(3)]
    bias = torch.nn.Parameter(torch.tensor([0.2]))
    y = torch.randint(0, 2, (B,)).float()
    loss, y_pred = mlp.bce_head(parts, bias, y)
    (2.0 * loss).backward()
    p2 = [p.detach().clone().requires_grad_(True) for p in parts]
    b2 = bias.detach().clone().requires_grad_(True)
    yp = torch.sigmoid(p2[0] + p2[1] + p2[2] + b2).squeeze(1)
    l2 = F.binary_cross_entropy(yp, y, reduction="sum")
    (2.0 * l2).backward()
    assert abs(loss.item() - l2.item()) < 1e-4 and torch.allclose(y_pred, yp.detach(), atol=1e-6)
    for a, b in zip(parts, p2):
        assert torch.allclose(a.grad, b.grad, atol=1e-5)
    assert torch.allclose(bias.grad, b2.grad, atol=1e-4)


def test_towers_wider_than_the_kernels_hold_stay_on_torch(mock):
    """Layers wider than csrc/mlp.hip's 16-sample LDS tiles hold (1152: the backward's two gradient tiles) must be
    declined by the layer spec (they run as nn.Linear on PyTorch-ROCm) instead of failing with ENOSUP at the first
    step; 1024-wide layers -- a common configuration -- are inside."""
    from deepctr_torch._hip import mlp
    from deepctr_torch.layers import DNN
    ok = DNN(40, (1024, 512, 256), device="cpu")
    wide = DNN(40, (1280, 64), device="cpu")
    lin_ok, lin_wide = torch.nn.Linear(256, 1, bias=False), torch.nn.Linear(64, 1, bias=False)
    assert mlp.tower_layers(ok, lin_ok) is not None and mlp.tower_layers(wide, lin_wide) is None
    # the widest accepted layer fits both budgets of the fused train step (150 KB): the forward with the smallest input
    # chunk (64 columns) and the backward's two [16, round64(w) + 4] gradient tiles
    w = mlp.MAX_TOWER_WIDTH
    assert 16 * (68 + 2 * ((w + 15) // 16 * 16 + 4)) * 4 <= 150 * 1024
    assert 16 * 2 * ((w + 63) // 64 * 64 + 4) * 4 <= 150 * 1024 < 16 * 2 * ((w + 64 + 63) // 64 * 64 + 4) * 4
    x = torch.randn(6, 40)
    y = mlp.tower(wide, lin_wide, x, 40)
    assert "mlp_fwd" not in mock.calls and torch.allclose(y, lin_wide(wide(x)), atol=1e-6)
