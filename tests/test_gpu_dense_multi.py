"""GPU: the multi-tensor optimizer step of the autograd-route train step (csrc/head.hip: dctr_dense_opt_multi,
dctr_l2_value_multi) against torch.optim on the same tensors, and the CIN pooling kernels (csrc/cin.hip:
dctr_cin_pool_fwd / _bwd) against the reference's split / sum(-1) formulation (interaction.py:226-246).

The optimizer kernel replaces ``torch.optim.SGD / Adagrad.step()`` (basemodel.py:262) and, for L2 terms, the part of
``get_regularization_loss()`` (basemodel.py:412-428) that reaches the gradients through autograd: both are elementwise
fp32 recurrences, so the comparison is (near) bit-level: rtol 2e-7 plus atol 1e-8 -- one ulp of the step term
lr * g / (sqrt(s) + eps) for parameters of magnitude ~1 (the kernel contracts s + g*g into an fma)."""
import ctypes

import numpy as np
import pytest
import torch

from np_oracle import optimizer_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _items(L, params, grads, states, l2s):
    items = (L.DenseItem * len(params))()
    for i, p in enumerate(params):
        items[i].p, items[i].g, items[i].n = p.data_ptr(), grads[i].data_ptr(), p.numel()
        items[i].state = states[i].data_ptr() if states is not None else None
        items[i].l2 = l2s[i]
    return items


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
@pytest.mark.parametrize("with_l2", [False, True])
def test_multi_tensor_step_matches_torch_optim(opt, with_l2):
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(3)
    # odd sizes, a tensor longer than one 4096-element chunk, 60 tensors (two launches of 48), misaligned views
    shapes = [(1,), (3,), (7, 5), (4097,), (128, 65), (1000,)] + [(17 + i,) for i in range(54)]
    base = [torch.randn(int(np.prod(s)) + 3, generator=g).to(DEV) for s in shapes]
    params = [b[1:1 + int(np.prod(s))].view(s) if i % 3 == 0 else b[:int(np.prod(s))].view(s)
              for i, (b, s) in enumerate(zip(base, shapes))]          # every third one starts 4 bytes off alignment
    l2s = [(1e-3 if (with_l2 and i % 2 == 0) else 0.0) for i in range(len(params))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    lr, eps = 0.01, 1e-10
    optim = torch.optim.SGD(ref, lr=lr) if opt == "sgd" else torch.optim.Adagrad(ref, lr=lr, eps=eps)
    states = [torch.zeros_like(p) for p in params] if opt == "adagrad" else None
    for step in range(3):
        grads = [torch.randn(p.shape, generator=g).to(DEV) * (10.0 ** (step - 1)) for p in params]
        # reference: the main gradient plus what autograd adds for lambda * sum(p^2), then the optimizer
        optim.zero_grad()
        reg = sum(torch.sum(l * torch.square(r)) for r, l in zip(ref, l2s) if l > 0) if with_l2 else None
        if reg is not None:
            reg.backward()
        for r, gr in zip(ref, grads):
            r.grad = gr.clone() if r.grad is None else r.grad + gr
        reg_ref = float(reg.detach()) if reg is not None else 0.0
        items = _items(L, params, grads, states, l2s)
        before = [(p.double().cpu().numpy().copy(), states[i].double().cpu().numpy().copy() if states is not None else None)
                  for i, p in enumerate(params)]
        if with_l2:
            out = torch.empty(1, device=DEV)
            L.check(lib.dctr_l2_value_multi(items, len(params), ctypes.c_void_p(out.data_ptr()), L.stream_handle(DEV)))
            assert abs(float(out) - reg_ref) <= 1e-5 * max(1.0, abs(reg_ref))
        optim.step()
        L.check(lib.dctr_dense_opt_multi(items, len(params), L.UPD_ADAGRAD if opt == "adagrad" else L.UPD_SGD, lr, eps,
                                         L.stream_handle(DEV)))
        torch.cuda.synchronize()
        for i, (p, r) in enumerate(zip(params, ref)):
            np.testing.assert_allclose(p.cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-7, atol=1e-8,
                                       err_msg="tensor %d step %d" % (i, step))
            # ... and against the oracle's statement of the same step in fp64 (np_oracle.optimizer_step: what
            # Oracle.train_step applies, pinned to the reference's 3-step goldens), the L2 term entering as 2 lambda p
            pw, sw = optimizer_step(opt, before[i][0], grads[i].double().cpu().numpy(), before[i][1], lr, eps, l2=l2s[i])
            np.testing.assert_allclose(p.double().cpu().numpy(), pw, rtol=2e-6, atol=2e-7, err_msg="oracle, tensor %d step %d" % (i, step))
            if opt == "adagrad":
                np.testing.assert_allclose(states[i].double().cpu().numpy(), sw, rtol=2e-6, atol=1e-9)
            if opt == "adagrad":
                # (s + g*g as one fma here, a product and a sum in torch: up to ~2 ulp)
                np.testing.assert_allclose(states[i].cpu().numpy(), optim.state[r]["sum"].cpu().numpy(), rtol=5e-7,
                                           atol=1e-10)
    for b, s in zip(base, shapes):       # nothing written outside the views
        assert torch.isfinite(b).all()


@pytest.mark.parametrize("B,O,D,nh", [(5, 8, 16, 4), (4096, 128, 16, 64), (33, 7, 5, 0), (2, 6, 16, 6)])
def test_cin_pool_kernels_match_split_sum(B, O, D, nh):
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    g = torch.Generator().manual_seed(B + O)
    A = torch.randn(B, O, D, generator=g).to(DEV)
    if nh < O:
        # the layer's block of a wider [B, ld] output: only its O - nh columns are written
        ld = O - nh + 5
        store = torch.full((B, ld), float("nan"), device=DEV)
        L.check(lib.dctr_cin_pool_fwd(P(A), B, O, D, nh, ctypes.c_void_p(store.data_ptr() + 8), ld, L.stream_handle(DEV)))
        ref = A[:, nh:].double().sum(-1)
        assert float((store[:, 2:2 + O - nh].double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        assert torch.isnan(store[:, :2]).all() and torch.isnan(store[:, 2 + O - nh:]).all()
    gh = torch.randn(B, nh, D, generator=g).to(DEV) if nh > 0 else None
    gp = torch.randn(B, O - nh, generator=g).to(DEV) if nh < O else None
    gA = torch.full((B, O, D), float("nan"), device=DEV)
    L.check(lib.dctr_cin_pool_bwd(P(gh), P(gp), O - nh, None, None, B, O, D, nh, nh, P(gA), L.stream_handle(DEV)))
    want = torch.zeros(B, O, D, device=DEV)
    if nh > 0:
        want[:, :nh] = gh
    if nh < O:
        want[:, nh:] = gp[:, :, None]
    assert torch.equal(gA, want)
    # with the layer's saved relu output: the relu's backward applied on the way (exact zeros, nothing else touched)
    gA.fill_(float("nan"))
    L.check(lib.dctr_cin_pool_bwd(P(gh), P(gp), O - nh, None, P(A), B, O, D, nh, nh, P(gA), L.stream_handle(DEV)))
    assert torch.equal(gA, torch.where(A > 0, want, torch.zeros_like(want)))
    # NULL inputs mean zero
    gA.fill_(float("nan"))
    L.check(lib.dctr_cin_pool_bwd(None, None, 0, None, None, B, O, D, nh, nh, P(gA), L.stream_handle(DEV)))
    assert torch.equal(gA, torch.zeros_like(gA))
    # without split_half every hidden row is a direct-connect row too (interaction.py:240-242): pool_from = 0, both terms
    gp_all = torch.randn(B, O + 3, generator=g).to(DEV)              # (rows O + 3 apart: a block of a wider gradient)
    gA.fill_(float("nan"))
    L.check(lib.dctr_cin_pool_bwd(P(gh), P(gp_all), O + 3, None, None, B, O, D, nh, 0, P(gA), L.stream_handle(DEV)))
    want = gp_all[:, :O, None].expand(B, O, D).clone()
    if nh > 0:
        want[:, :nh] = gh + want[:, :nh]      # (the kernel's order: hidden term first)
    assert torch.equal(gA, want)
    # the 1-unit projection's backward folded in (xdeepfm.py:72): gp(b, j) = g_logit[b] * w_head[j], one product in fp32
    if nh < O:
        g_logit = torch.randn(B, generator=g).to(DEV)
        w_head = torch.randn(O - nh, generator=g).to(DEV)
        gA.fill_(float("nan"))
        L.check(lib.dctr_cin_pool_bwd(P(gh), P(g_logit), 1, P(w_head), P(A), B, O, D, nh, nh, P(gA), L.stream_handle(DEV)))
        want = torch.zeros(B, O, D, device=DEV)
        if nh > 0:
            want[:, :nh] = gh
        want[:, nh:] = (g_logit[:, None] * w_head[None, :])[:, :, None]
        assert torch.equal(gA, torch.where(A > 0, want, torch.zeros_like(want)))


@pytest.mark.parametrize("B,N,ld", [(1, 1, 1), (5, 192, 192), (4096, 192, 200), (33, 70, 71), (257, 1000, 1000)])
def test_rows_tdot_matches_fp64(B, N, ld):
    """dctr_rows_tdot: out[j] = sum_b w[b] x[b, j] (the weight gradient of xDeepFM's cin_linear, xdeepfm.py:72: g^T X) --
    against fp64 and torch.mm's own result, bit-identical run to run (fixed summation order)."""
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(7 * B + N)
    x = torch.randn(B, ld, generator=g).to(DEV)
    w = torch.randn(B, generator=g).to(DEV)
    ws = torch.empty((max(1, lib.dctr_relu_bwd_bias_workspace_floats(B, N)),), device=DEV)
    outs = []
    for _ in range(2):
        out = torch.full((N,), float("nan"), device=DEV)
        ws.fill_(float("nan"))
        L.check(lib.dctr_rows_tdot(ctypes.c_void_p(x.data_ptr()), ld, ctypes.c_void_p(w.data_ptr()), B, N,
                                   ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), L.stream_handle(DEV)))
        outs.append(out)
    ref = w.double() @ x[:, :N].double()
    scale = float((w.double().abs() @ x[:, :N].double().abs()).max())
    assert float((outs[0].double() - ref).abs().max()) <= 1e-6 * max(1.0, scale)
    assert float((outs[0] - torch.mm(w.reshape(1, B), x[:, :N]).reshape(-1)).abs().max()) <= 2e-6 * max(1.0, scale)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B,N,ld", [(1, 1, 1), (5, 192, 192), (4096, 192, 200), (33, 70, 71), (257, 1000, 1000)])
def test_rows_dot_matches_fp64(B, N, ld):
    """dctr_rows_dot: out[b] = sum_j x[b, j] w[j] (nn.Linear(N, 1, bias=False), xdeepfm.py:72) -- against fp64, and
    bit-identical run to run (fixed summation order)."""
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(B + N)
    x = torch.randn(B, ld, generator=g).to(DEV)
    w = torch.randn(N, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        out = torch.full((B,), float("nan"), device=DEV)
        L.check(lib.dctr_rows_dot(ctypes.c_void_p(x.data_ptr()), ld, ctypes.c_void_p(w.data_ptr()), B, N,
                                  ctypes.c_void_p(out.data_ptr()), L.stream_handle(DEV)))
        outs.append(out)
    ref = x[:, :N].double() @ w.double()
    scale = float((x[:, :N].double().abs() @ w.double().abs()).max())
    assert float((outs[0].double() - ref).abs().max()) <= 1e-6 * max(1.0, scale)
    assert torch.equal(outs[0], outs[1])
