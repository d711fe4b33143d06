"""GPU: the autograd-glue launches of csrc/head.hip (dctr_rows_join, dctr_relu_bwd_bias) against the torch ops they replace
-- the slice backward of the gather's output (two copies + a fill), aten::threshold_backward + sum(0) behind a wide
nn.Linear (reference layers/core.py:120-134) -- and through the autograd Functions that call them."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,W,nd,ld", [(4096, 416, 13, 432), (77, 64, 0, 64), (5, 8, 3, 12), (1, 16, 1, 20)])
@pytest.mark.parametrize("with_c", [False, True])
def test_rows_join(B, W, nd, ld, with_c):
    from deepctr_torch._hip import lib as L
    g = torch.Generator(device=DEV).manual_seed(B + W)
    a = torch.randn(B, W, device=DEV, generator=g)
    c = torch.randn(B, W + 4, device=DEV, generator=g)[:, :W] if with_c else None      # (a strided second addend)
    d = torch.randn(B, nd + 2, device=DEV, generator=g)[:, 1:1 + nd] if nd else None     # (an unaligned dense view)
    out = torch.full((B, ld), float("nan"), device=DEV)
    L.check(L.lib().dctr_rows_join(_p(a), a.stride(0), _p(c), c.stride(0) if with_c else 0, W, _p(d),
                                   d.stride(0) if nd else 0, nd, _p(out), ld, B, L.stream_handle(torch.device(DEV))),
            "dctr_rows_join")
    want = torch.zeros(B, ld, device=DEV)
    want[:, :W] = a + c if with_c else a
    if nd:
        want[:, W:W + nd] = d
    assert torch.equal(out, want)


@pytest.mark.parametrize("B,N", [(4096, 128), (100, 300), (33, 7), (1, 1)])
@pytest.mark.parametrize("relu", [True, False])
def test_relu_bwd_bias(B, N, relu):
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    g = torch.Generator(device=DEV).manual_seed(B * 7 + N)
    gr = torch.randn(B, N, device=DEV, generator=g)
    h = torch.relu(torch.randn(B, N, device=DEV, generator=g)) if relu else None
    go = torch.empty(B, N, device=DEV) if relu else None
    gb = torch.empty(N, device=DEV)
    ws = torch.empty(max(1, lib.dctr_relu_bwd_bias_workspace_floats(B, N)), device=DEV)
    L.check(lib.dctr_relu_bwd_bias(_p(gr), gr.stride(0), _p(h), N if relu else 0, B, N, _p(go), N, _p(gb), _p(ws),
                                   L.stream_handle(torch.device(DEV))), "dctr_relu_bwd_bias")
    want = torch.ops.aten.threshold_backward(gr, h, 0) if relu else gr
    if relu:
        assert torch.equal(go, want)
    # the oracle's statement of the same two autograd nodes (np_oracle.relu_bias_backward: what dnn_backward is made of)
    from np_oracle import relu_bias_backward
    gz64, gb64 = relu_bias_backward(gr.double().cpu().numpy(), h.double().cpu().numpy() if relu else None)
    if relu:
        assert float((go.double().cpu() - torch.from_numpy(gz64)).abs().max()) == 0.0
    ref = torch.from_numpy(gb64).to(DEV)
    assert float((gb.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    gb2 = torch.empty(N, device=DEV)
    L.check(lib.dctr_relu_bwd_bias(_p(gr), gr.stride(0), _p(h), N if relu else 0, B, N, _p(go), N, _p(gb2), _p(ws),
                                   L.stream_handle(torch.device(DEV))), "dctr_relu_bwd_bias")
    assert torch.equal(gb, gb2)            # fixed order of additions


def test_wide_linear_and_split_functions_match_plain_autograd(monkeypatch):
    """WideLinearFunction / SplitGatheredFunction with the glue launches against the same Functions without them."""
    from deepctr_torch._hip import mlp as _mlp
    from deepctr_torch._hip import ops as _ops
    gen = torch.Generator(device=DEV).manual_seed(3)
    B, K, N = 512, 5000, 128
    x0 = torch.randn(B, K, device=DEV, generator=gen) * 0.1
    W0 = torch.randn(N, K, device=DEV, generator=gen) * 0.02
    b0 = torch.randn(N, device=DEV, generator=gen) * 0.1
    r = torch.randn(B, N, device=DEV, generator=gen)
    full0 = torch.randn(B, 432, device=DEV, generator=gen)
    r_e, r_d = torch.randn(B, 26, 16, device=DEV, generator=gen), torch.randn(B, 13, device=DEV, generator=gen)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DCTR_GLUE_KERNELS", mode)
        monkeypatch.setenv("DCTR_TUNABLE_GEMM", "0")
        x, W, b = x0.clone().requires_grad_(), W0.clone().requires_grad_(), b0.clone().requires_grad_()
        (_mlp.WideLinearFunction.apply(x, W, b, True) * r).sum().backward()
        full = full0.clone().requires_grad_()
        emb, dense = _ops.SplitGatheredFunction.apply(full, 416, 26, 16, 13)
        ((emb * r_e).sum() + (dense * r_d).sum()).backward()
        got[mode] = (x.grad, W.grad, b.grad, full.grad)
    for a, c in zip(got["1"][:2], got["0"][:2]):
        assert torch.equal(a, c)
    assert float((got["1"][2] - got["0"][2]).abs().max()) <= 1e-5 * float(got["0"][2].abs().max())
    assert torch.equal(got["1"][3], got["0"][3])


def test_stamp_writes_an_increasing_device_clock_also_inside_a_graph():
    """dctr_stamp (csrc/sync.hip; what bench.py times launches with INSIDE hipGraph replays): a one-lane kernel that stores
    the device's real-time counter.  Stamps of one stream are ordered; the distance of two stamps around a launch covers the
    launch; replaying the captured sequence refreshes them."""
    import ctypes
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    st = torch.zeros(4, dtype=torch.int64, device=DEV)
    x = torch.randn(1 << 22, device=DEV)

    def seq():
        h = L.stream_handle(DEV)
        L.check(lib.dctr_stamp(ctypes.c_void_p(st.data_ptr()), h), "dctr_stamp")
        L.check(lib.dctr_stamp(ctypes.c_void_p(st.data_ptr() + 8), h), "dctr_stamp")
        x.mul_(1.0001)
        L.check(lib.dctr_stamp(ctypes.c_void_p(st.data_ptr() + 16), h), "dctr_stamp")

    seq()
    torch.cuda.synchronize()
    a = st.cpu().tolist()
    assert 0 < a[0] <= a[1] < a[2], a
    assert a[2] - a[1] > a[1] - a[0] >= 0          # (an 16 MB elementwise launch sits between the last two)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        seq()
        with torch.cuda.graph(g, stream=s):
            seq()
    g.replay()
    torch.cuda.synchronize()
    b = st.cpu().tolist()
    assert b[0] > a[2] and b[0] <= b[1] < b[2], (a, b)
