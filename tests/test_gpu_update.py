"""GPU: the deterministic fused backward + optimizer kernel (csrc/update.hip) against a plain PyTorch fp32
reference of the same op (index_add of autograd gradients), over duplicate-heavy id distributions, odd batch
sizes, every lane layout (dim 16 -> float4 x 4 lanes, dim 6 -> float2, dim 5 -> scalar) and all three
update modes.  Also: bit-reproducibility (the reason the kernel exists) and untouched-row invariance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(n_sparse, vocabs, dim, n_dense, linear=True):
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("s%d" % i, vocabs[i], dim) for i in range(n_sparse)] + \
           [DenseFeat("d%d" % i, 1) for i in range(n_dense)]
    m = DeepFM(cols if linear else [], cols, dnn_hidden_units=(8,), l2_reg_linear=0, l2_reg_embedding=0,
               init_std=0.1, device=DEV)
    return m


def _batch(B, vocabs, n_dense, mode, seed):
    g = torch.Generator().manual_seed(seed)
    cols = []
    for v in vocabs:
        if mode == "uniform":
            ids = torch.randint(0, v, (B,), generator=g)
        elif mode == "hot":          # one id takes ~90 % of the batch: long segments spanning many tiles
            ids = torch.where(torch.rand(B, generator=g) < 0.9, torch.full((B,), min(3, v - 1)),
                              torch.randint(0, v, (B,), generator=g))
        elif mode == "zipf":         # Zipf(1.05) ranks: a few hot ids of a few hundred samples each, a long tail --
            r = torch.arange(1, v + 1, dtype=torch.float64)   # partitions of 300-500 entries with multi-tile segments
            ids = torch.multinomial(r.pow(-1.05), B, replacement=True, generator=g)
        else:                        # "same": every sample hits one row
            ids = torch.full((B,), v - 1)
        cols.append(ids.float())
    X = torch.stack(cols, 1)
    if n_dense:
        X = torch.cat([X, torch.rand(B, n_dense, generator=g)], 1)
    return X.to(DEV)


def _reference_grads(m, X, R_out, r_wide, r_fm, want_fm):
    """Dense [V, D] gradients of  sum(out*R_out) + sum(wide*r_wide) + sum(fm*r_fm)  by autograd on plain torch ops."""
    plan = m.model_plan()
    names = [f.name for f in plan.deep]
    deep = [m.embedding_dict[n].weight.detach().clone().requires_grad_(True) for n in names]
    wide = [m.linear_model.embedding_dict[f.name].weight.detach().clone().requires_grad_(True) for f in plan.wide]
    ids = X[:, :len(names)].long()
    E = torch.stack([deep[i][ids[:, i]] for i in range(len(names))], 1)          # [B, F, D]
    loss = (E.reshape(X.shape[0], -1) * R_out[:, :E.shape[1] * E.shape[2]]).sum()
    if want_fm:
        fm = 0.5 * (E.sum(1).pow(2) - E.pow(2).sum(1)).sum(1)
        loss = loss + (fm * r_fm).sum()
    if wide:
        wsum = sum(wide[i][X[:, plan.wide[i].col].long(), 0] for i in range(len(wide)))
        loss = loss + (wsum * r_wide).sum()
    grads = torch.autograd.grad(loss, deep + wide)
    return names, deep, wide, grads[:len(deep)], grads[len(deep):]


CASES = [
    # (B, vocabs, dim, n_dense, id mode)
    (1, [7, 5, 9], 16, 2, "uniform"),
    (63, [7, 5, 9], 16, 0, "uniform"),
    (65, [50, 3, 1000], 16, 1, "hot"),
    (300, [40, 40, 40, 11], 6, 3, "uniform"),
    (777, [13, 1000, 2], 5, 0, "hot"),
    (4096, [100_000, 17, 1_000_000, 3], 16, 13, "uniform"),
    (4096, [100_000, 17], 16, 0, "same"),
    (5000, [1000, 50_000], 12, 1, "hot"),
    (4096, [100_000, 1000, 40], 16, 1, "zipf"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_D%d_%s" % (c[0], c[2], c[4]))
@pytest.mark.parametrize("mode", ["dense", "sgd", "adagrad"])
def test_update_kernel_matches_torch(case, mode):
    B, vocabs, dim, n_dense, idmode = case
    m = _model(len(vocabs), vocabs, dim, n_dense)
    plan = m.model_plan()
    X = _batch(B, vocabs, n_dense, idmode, seed=B)
    assert plan.unit_path
    plan.bind(X.device)
    assert plan.update_kernel_ok(B)
    gen = torch.Generator(device=DEV).manual_seed(1)
    R_out = torch.randn(B, plan.width, device=DEV, generator=gen)
    r_wide = torch.randn(B, device=DEV, generator=gen)
    r_fm = torch.randn(B, device=DEV, generator=gen)
    names, deep0, wide0, g_deep, g_wide = _reference_grads(m, X, R_out, r_wide, r_fm, True)

    lr, eps = 0.05, 1e-10
    if mode == "sgd":
        m.compile(torch.optim.SGD(m.parameters(), lr=lr), "binary_crossentropy")
    elif mode == "adagrad":
        m.compile(torch.optim.Adagrad(m.parameters(), lr=lr), "binary_crossentropy")
        for p in plan.table_params:      # non-trivial starting state
            m.optim.state[p]["sum"].uniform_(0.0, 0.5)
        state0 = {id(p): m.optim.state[p]["sum"].clone() for p in plan.table_params}
    assert plan.update[0] == mode

    out, wide, fm = m.fused_inputs(X, want_fm=True)
    loss = (out * R_out).sum() + (wide.squeeze(1) * r_wide).sum() + (fm.squeeze(1) * r_fm).sum()
    loss.backward()
    torch.cuda.synchronize()
    plan.check_ids()

    def close(a, b, what):
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert err <= 3e-5 * scale, "%s: max|d|=%.3e (scale %.3g)" % (what, err, scale)

    for i, n in enumerate(names):
        p = m.embedding_dict[n].weight
        if mode == "dense":
            close(p.grad, g_deep[i], "grad " + n)
        elif mode == "sgd":
            close(p.detach(), deep0[i].detach() - lr * g_deep[i], "table " + n)
        else:
            s = state0[id(p)] + g_deep[i] * g_deep[i]
            close(m.optim.state[p]["sum"], s, "state " + n)
            close(p.detach(), deep0[i].detach() - lr * g_deep[i] / (s.sqrt() + eps), "table " + n)
            untouched = g_deep[i].abs().sum(1) == 0
            assert torch.equal(p.detach()[untouched], deep0[i].detach()[untouched])
    for i, f in enumerate(plan.wide):
        p = m.linear_model.embedding_dict[f.name].weight
        if mode == "dense":
            close(p.grad, g_wide[i], "wide grad " + f.name)
        elif mode == "sgd":
            close(p.detach(), wide0[i].detach() - lr * g_wide[i], "wide table " + f.name)
        else:
            s = state0[id(p)] + g_wide[i] * g_wide[i]
            close(p.detach(), wide0[i].detach() - lr * g_wide[i] / (s.sqrt() + eps), "wide table " + f.name)


def test_update_kernel_is_bit_reproducible():
    """Same inputs -> bit-identical tables, whatever the workgroup schedule (no atomics anywhere)."""
    B, vocabs = 4096, [1000, 17, 100_000, 3]
    X = _batch(B, vocabs, 2, "hot", seed=5)
    results = []
    for rep in range(3):
        torch.manual_seed(0)
        m = _model(len(vocabs), vocabs, 16, 2)
        m.compile("adagrad", "binary_crossentropy")
        gen = torch.Generator(device=DEV).manual_seed(1)
        R_out = torch.randn(B, m.model_plan().width, device=DEV, generator=gen)
        for _ in range(2):
            out, wide, fm = m.fused_inputs(X, want_fm=True)
            ((out * R_out).sum() + wide.sum() + fm.sum()).backward()
        torch.cuda.synchronize()
        results.append([p.detach().clone() for p in m.model_plan().table_params])
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert torch.equal(a, b)


def test_wide_only_and_deep_only_units():
    """linear_feature_columns != dnn_feature_columns: unpaired units."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    a, b, c = SparseFeat("a", 50, 8), SparseFeat("b", 60, 8), SparseFeat("c", 70, 8)
    m = DeepFM([a, c, DenseFeat("x", 1)], [a, b], dnn_hidden_units=(4,), l2_reg_linear=0, l2_reg_embedding=0,
               init_std=0.1, device=DEV)
    plan = m.model_plan()
    assert plan.unit_path and sorted(u[:2] for u in plan.units) == [(-1, 1), (0, 0), (1, -1)]
    B = 200
    g = torch.Generator().manual_seed(0)
    X = torch.stack([torch.randint(0, 50, (B,), generator=g).float(), torch.randint(0, 70, (B,), generator=g).float(),
                     torch.rand(B, generator=g), torch.randint(0, 60, (B,), generator=g).float()], 1).to(DEV)
    assert list(m.feature_index) == ["a", "c", "x", "b"]
    y = torch.randint(0, 2, (B,), generator=g).float().to(DEV)
    ref = {k: v.detach().clone() for k, v in m.state_dict().items()}
    yp = m(X).squeeze()
    torch.nn.functional.binary_cross_entropy(yp, y, reduction="sum").backward()
    # torch reference
    P = {k: v.clone().requires_grad_(True) for k, v in ref.items()}
    ia, ic, ib = X[:, 0].long(), X[:, 1].long(), X[:, 3].long()
    E = torch.stack([P["embedding_dict.a.weight"][ia], P["embedding_dict.b.weight"][ib]], 1)
    lin = P["linear_model.embedding_dict.a.weight"][ia] + P["linear_model.embedding_dict.c.weight"][ic] + \
        X[:, 2:3] @ P["linear_model.weight"]
    fm = 0.5 * (E.sum(1).pow(2) - E.pow(2).sum(1)).sum(1, keepdim=True)
    h = torch.relu(E.reshape(B, -1) @ P["dnn.linears.0.weight"].t() + P["dnn.linears.0.bias"])
    logit = lin + fm + h @ P["dnn_linear.weight"].t() + P["out.bias"]
    torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit).squeeze(), y, reduction="sum").backward()
    for k, p in m.named_parameters():
        assert float((p.grad - P[k].grad).abs().max()) <= 2e-5 * max(1.0, float(P[k].grad.abs().max())), k


@pytest.mark.parametrize("idmode,B", [("uniform", 4096), ("hot", 4096), ("same", 777), ("hot", 20000),
                                          ("zipf", 4096), ("zipf", 16384),
                                          # (the two-level pre-pass of large batches: k_prepass_bin + k_prepass_sort)
                                          ("uniform", 70000), ("zipf", 40000), ("same", 9000)])
def test_prepass_layouts_and_inkernel_scan_agree_bit_for_bit(monkeypatch, idmode, B):
    """The same two Adagrad steps five ways -- (a) segment pre-pass (dctr_embed_segments) + interleaved slabs, the
    default; (b) no separate pre-pass (large batches: the update buckets / pre-sorts in line); (c) the reference's
    contiguous tensors; (d) the block layout; (e) no workspace: every workgroup scans and sorts for itself --
    must leave bit-identical tables and optimizer state: the layout only moves bytes, the pre-pass only moves work."""
    vocabs = [1000, 17, 100_000, 3]
    X = _batch(B, vocabs, 2, idmode, seed=11)
    results = []
    # (bucket "0": no workspace at all -- every workgroup of the general kernel scans the unit's ids and sorts for itself;
    # from B = 16 384 the other variants all pass through the two-level pre-pass, this one never does)
    for seg, layout, bucket in (("1", "interleaved", "auto"), ("0", "interleaved", "auto"), ("1", "contiguous", "auto"),
                                ("1", "block", "auto"), ("0", "interleaved", "0")):
        monkeypatch.setenv("DCTR_SEGMENTS", seg)
        monkeypatch.setenv("DCTR_TABLE_LAYOUT", layout)
        monkeypatch.setenv("DCTR_UPD_BUCKET", bucket)
        torch.manual_seed(0)
        m = _model(len(vocabs), vocabs, 16, 2)
        m.compile("adagrad", "binary_crossentropy")
        plan = m.model_plan()
        p0 = plan.table_params[0]
        assert p0.stride(0) == {"interleaved": 32, "contiguous": 16, "block": 64}[layout]
        gen = torch.Generator(device=DEV).manual_seed(1)
        R_out = torch.randn(B, plan.width, device=DEV, generator=gen)
        for _ in range(2):
            out, wide, fm = m.fused_inputs(X, want_fm=True)
            ((out * R_out).sum() + wide.sum() + fm.sum()).backward()
        torch.cuda.synchronize()
        plan.check_ids()
        results.append([p.detach().clone().contiguous() for p in plan.table_params] +
                       [m.optim.state[p]["sum"].clone().contiguous() for p in plan.table_params])
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("idmode,B", [("uniform", 4096), ("zipf", 20000), ("hot", 40000)])
def test_prepass_buckets_equal_the_numpy_statement(idmode, B):
    """dctr_embed_ids + dctr_embed_segments against oracle/prepass_oracle.segments_direct: every (unit, partition)'s count and
    its sorted keys, for the tag-scan path (B < 16 384) and the two-level path (k_prepass_bin + k_prepass_sort) alike."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from prepass_oracle import segments_direct
    from deepctr_torch._hip import lib as L
    vocabs = [1000, 17, 100_000]
    X = _batch(B, vocabs, 1, idmode, seed=23)
    m = _model(len(vocabs), vocabs, 16, 1)
    m.compile("adagrad", "binary_crossentropy")
    plan = m.model_plan()
    dev = torch.device(DEV)
    cplan = plan.bind(dev)
    lib = L.lib()
    nu = len(plan.units)
    P = int(lib.dctr_embed_update_partitions(cplan, B))
    n_ws = int(lib.dctr_embed_update_workspace_ints(cplan, nu, B))
    ws = torch.zeros(n_ws, dtype=torch.int32, device=dev)
    ids_t = torch.empty(nu, B, dtype=torch.int32, device=dev)
    parts_t = torch.empty(nu, B, dtype=torch.int16, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    s = L.stream_handle(dev)
    L.check(lib.dctr_embed_ids(cplan, plan.units_ptr(), nu, p(X), X.stride(0), B, p(ids_t), p(parts_t), s), "ids")
    L.check(lib.dctr_embed_segments(cplan, plan.units_ptr(), nu, plan.max_vocab, p(ids_t), p(parts_t), B, p(ws), n_ws, s),
            "segments")
    torch.cuda.synchronize()
    cnt = ws[:nu * P].cpu().numpy().reshape(nu, P)
    keys = ws[nu * P:nu * P * 513].cpu().numpy().view(np.uint32).reshape(nu, P, 512)
    Xc = X.cpu().numpy()
    for u, (di, wi, col, _) in enumerate(plan.units):
        vocab = plan.deep[di].vocab if di >= 0 else plan.wide[wi].vocab
        c0, k0 = segments_direct(Xc[:, col].astype(np.int64), vocab, P)
        assert np.array_equal(cnt[u], c0), u
        for q in range(P):
            if c0[q] <= 512:
                assert np.array_equal(keys[u, q, :c0[q]], k0[q]), (u, q)
