"""The round-4 train step (deepctr_torch/_hip/step.py: the embedding lookup inside the tower launch,
dctr_embed_tower_train_step, five launches enqueued straight through the C ABI) against the autograd-assembled fused
step of rounds 1-3 (dctr_embed_fwd + dctr_mlp_train_step + ...; DCTR_STEP_ENGINE=0).  Same arithmetic in the same order:
everything observable must be bit-identical -- the kernel's outputs, the losses, every parameter and every optimizer
state after many steps, eager and graph-replayed.  The older path is itself pinned to the reference's goldens
(tests/test_gpu_deepfm.py, test_gpu_full_golden.py), and the full-size goldens run through the new one by default.

Small vocabularies on purpose: the gather of step n + 1 reads rows the update of step n wrote microseconds earlier."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# pooled = (maxlen of a mean history over its own table [ids != 0 mask], maxlen of a sum history that shares C1's table and
# has a length column): the general update units of round 5 -- pooled VarLen fields, a shared table -- inside the engine
def _cols(vocab, F, D, n_dense, pooled=None):
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    cols = [SparseFeat("C%d" % (i + 1), vocab, D) for i in range(F)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(n_dense)]
    if pooled:
        cols.append(VarLenSparseFeat(SparseFeat("hist", vocab, D), maxlen=pooled[0], combiner="mean"))
        cols.append(VarLenSparseFeat(SparseFeat("seq", vocab, D, embedding_name="C1"), maxlen=pooled[1], combiner="sum",
                                     length_name="seq_len"))
        if len(pooled) > 2:      # a max-pooled history (ids != 0 mask; at least one position valid: tests/matrix_data.py clean_rows)
            cols.append(VarLenSparseFeat(SparseFeat("kw", vocab, D), maxlen=pooled[2], combiner="max"))
    return cols


def _model(kind, vocab, opt, F=26, D=16, n_dense=13, hidden=(256, 128), pooled=None):
    from deepctr_torch import models as M
    cols = _cols(vocab, F, D, n_dense, pooled)
    kw = dict(dnn_hidden_units=hidden, l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0, seed=1024, device=DEV)
    m = M.DeepFM(cols, cols, **kw) if kind == "deepfm" else M.WDL(cols, cols, **kw)
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()
    return m


def _data(vocab, F, n_dense, B, n_batches, seed=7, pooled=None):
    gen = torch.Generator().manual_seed(seed)
    n = B * n_batches
    ids = torch.randint(0, vocab, (n, F), generator=gen)
    X = torch.cat([ids.float(), torch.rand(n, n_dense, generator=gen)], dim=1)
    if pooled:
        T0, T1 = pooled[0], pooled[1]
        h = torch.randint(1, vocab, (n, T0), generator=gen)
        h = h * (torch.arange(T0)[None, :] < torch.randint(0, T0 + 1, (n, 1), generator=gen))     # 0-padded, some empty
        sq = torch.randint(0, vocab, (n, T1), generator=gen)
        ln = torch.randint(0, T1 + 1, (n, 1), generator=gen)
        X = torch.cat([X, h.float(), sq.float(), ln.float()], dim=1)      # (inputs.py:99-123: positions, then the length)
        if len(pooled) > 2:
            T2 = pooled[2]
            kw = torch.randint(1, vocab, (n, T2), generator=gen)
            kw = kw * (torch.arange(T2)[None, :] < torch.randint(1, T2 + 1, (n, 1), generator=gen))
            X = torch.cat([X, kw.float()], dim=1)
    X = X.to(DEV)
    y = torch.randint(0, 2, (n,), generator=gen).float().to(DEV)
    return X, y


def _run(engine, kind, vocab, opt, steps, graphed, B=4096, F=26, D=16, n_dense=13, hidden=(256, 128), topo=None,
         pooled=None):
    os.environ["DCTR_STEP_ENGINE"] = "1" if engine else "0"
    if topo:
        os.environ["DCTR_STEP_TOPOLOGY"] = topo
    try:
        m = _model(kind, vocab, opt, F, D, n_dense, hidden, pooled)
        X, y = _data(vocab, F, n_dense, B, 8, pooled=pooled)
        bat = lambda i: (X[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B])   # noqa: E731
        losses, preds = [], []
        i = 0
        for _ in range(2):
            out = m._train_step(*bat(i))
            losses.append(out[0].clone())
            preds.append(out[2].clone())
            i += 1
        st = m._fused_step_state()
        assert st is not None
        used = st.get("engine") is not None and st["engine"].supports(*bat(0))
        assert used == engine, "engine %s but used=%s" % (engine, used)
        if graphed:
            from deepctr_torch._hip.graph import GraphedTrainStep
            S = 4
            g = GraphedTrainStep(m, *bat(0), steps_per_graph=S, inputs_ready=True).capture(*bat(0))
            while i < steps:
                for _ in range(S):
                    out = g(*bat(i))
                    i += 1
                g.flush()
                torch.cuda.synchronize()
                losses.append(out[0].clone())
                preds.append(out[2].clone())
        else:
            while i < steps:
                out = m._train_step(*bat(i))
                losses.append(out[0].clone())
                i += 1
            preds.append(out[2].clone())
        torch.cuda.synchronize()
        m.model_plan().check_ids()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        ost = {}
        names = {id(p): n for n, p in m.named_parameters()}
        for grp in m.optim.param_groups:
            for p in grp["params"]:
                for k, v in m.optim.state.get(p, {}).items():
                    if torch.is_tensor(v) and v.numel() > 1:
                        ost["%s/%s" % (names.get(id(p), "?"), k)] = v.detach().clone()
        return sd, ost, torch.stack([l.reshape(()) for l in losses]).cpu(), [p.cpu() for p in preds]
    finally:
        os.environ.pop("DCTR_STEP_ENGINE", None)
        os.environ.pop("DCTR_STEP_TOPOLOGY", None)


def _same(a, b, what):
    ref_sd, ref_st, ref_loss, ref_pred = a
    sd, st, loss, pred = b
    assert torch.equal(loss, ref_loss), "%s: losses differ, first at entry %d: %r vs %r" % (
        what, int((loss != ref_loss).nonzero()[0]), loss[:4].tolist(), ref_loss[:4].tolist())
    for p, q in zip(pred, ref_pred):
        assert torch.equal(p, q), "%s: predictions differ (max %.3e)" % (what, float((p - q).abs().max()))
    for k in ref_sd:
        assert torch.equal(sd[k], ref_sd[k]), "%s: %s differs (max %.3e)" % (what, k, float((sd[k] - ref_sd[k]).abs().max()))
    assert set(st) == set(ref_st)
    for k in ref_st:
        assert torch.equal(st[k], ref_st[k]), "%s: optimizer state %s differs" % (what, k)


@pytest.mark.parametrize("graphed", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("opt", ["adagrad", "sgd"])
def test_engine_leaves_the_same_bits_as_the_two_launch_step(opt, graphed):
    _same(_run(False, "deepfm", 3000, opt, 42, graphed), _run(True, "deepfm", 3000, opt, 42, graphed), "deepfm/" + opt)


@pytest.mark.parametrize("case", [
    dict(kind="wdl", D=16),                                   # no FM part
    dict(kind="deepfm", D=8, F=5, n_dense=3, hidden=(64, 32), B=1000),       # 2 lanes per row, ragged last tile (1000 = 62.5 x 16)
    dict(kind="deepfm", D=32, F=7, n_dense=0, hidden=(128,), B=512),         # 8 lanes per row, no dense block
    dict(kind="deepfm", D=4, F=30, n_dense=20, hidden=(128, 64), B=256),     # 1 lane per row, > 16 wide fields per pass
    # pooled VarLen fields + a shared table inside the fused gather (general update units)
    dict(kind="deepfm", pooled=(8, 5)),
    dict(kind="deepfm", pooled=(4, 3, 5)),                                   # + a max-pooled history: arg-max from the tower launch
    dict(kind="wdl", D=8, F=6, n_dense=2, hidden=(128, 64), B=777, pooled=(3, 2, 4)),
    dict(kind="wdl", D=8, F=5, n_dense=3, hidden=(64, 32), B=1000, pooled=(3, 2)),
    dict(kind="deepfm", D=32, F=4, n_dense=0, hidden=(256, 128), B=512, pooled=(6, 4)),   # (the positions' rows are staged in
    # the backward's LDS image: a narrow tower leaves no room and keeps the two-launch step)
])
def test_engine_shapes(case):
    kw = dict(F=26, D=16, n_dense=13, hidden=(256, 128), B=4096)
    kw.update(case)
    kind = kw.pop("kind")
    _same(_run(False, kind, 500, "adagrad", 10, False, **kw), _run(True, kind, 500, "adagrad", 10, False, **kw), repr(case))


def test_engine_runs_pooled_fields_graph_replayed():
    """26 + 13 Criteo columns + a mean history + a sum history over C1's table: 42 graph-replayed steps, bit for bit the
    two-launch step (whose pooled lookup and sorted update are pinned to the reference's goldens)."""
    _same(_run(False, "deepfm", 3000, "adagrad", 42, True, pooled=(8, 5)),
          _run(True, "deepfm", 3000, "adagrad", 42, True, pooled=(8, 5)), "deepfm + pooled")


@pytest.mark.parametrize("topo", ["serial", "weights_flag"])
def test_engine_topologies(topo):
    """Every way of enqueueing the step leaves the default's bits.  Round 6: ``weights_flag`` (tower -> update on one queue,
    the dense parameters handed to the next tower launch through DCTR_SYNC_W_GEN, the reduction folded into the
    weight-gradient launch: _hip/step.py) -- opt-in, slower than the default, kept exact."""
    ref = _run(True, "deepfm", 3000, "adagrad", 42, True)
    _same(ref, _run(True, "deepfm", 3000, "adagrad", 42, True, topo=topo), topo)
    if topo == "weights_flag":
        _same(_run(True, "deepfm", 3000, "sgd", 22, True, pooled=(4, 3, 5)),
              _run(True, "deepfm", 3000, "sgd", 22, True, topo=topo, pooled=(4, 3, 5)), topo + " + pooled")
        _same(_run(True, "wdl", 3000, "sgd", 6, False), _run(True, "wdl", 3000, "sgd", 6, False, topo=topo), topo + " eager")


def test_engine_long_run_on_large_tables():
    """1M-row tables (the benchmark shape), 200 graph-replayed steps: rows mostly miss every cache"""
    _same(_run(False, "deepfm", 1_000_000, "adagrad", 202, True), _run(True, "deepfm", 1_000_000, "adagrad", 202, True),
          "1M rows")


def test_out_of_range_id_is_flagged():
    m = _model("deepfm", 100, "adagrad")
    X, y = _data(100, 26, 13, 256, 1)
    X[5, 3] = 100.0
    m._train_step(X, y)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        m.model_plan().check_ids()


def test_kernel_outputs_equal_the_two_launches():
    """dctr_embed_tower_train_step against dctr_embed_fwd + dctr_mlp_train_step(defer_wgrad) on the same inputs: the gathered
    rows, sum_f e, predictions, d loss / d logit and d loss / d input, bit for bit."""
    from deepctr_torch._hip import lib as L
    from deepctr_torch._hip import step as S
    lib = L.lib()
    B = 4096
    m = _model("deepfm", 50_000, "adagrad")
    X, y = _data(50_000, 26, 13, B, 1)
    m._train_step(X, y)                      # builds the slab, the plan and the engine's buffers
    torch.cuda.synchronize()
    st = m._fused_step_state()
    eng = st["engine"]
    b = eng._buffers(B, X.device)
    plan = m.model_plan()
    cplan = plan.bind(X.device)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
    s = L.stream_handle(X.device)
    f32 = dict(dtype=torch.float32, device=DEV)
    out0, wide0, fm0 = torch.zeros(B, plan.ld_out, **f32), torch.empty(B, **f32), torch.empty(B, **f32)
    fms0 = torch.empty(B, 16, **f32)
    L.check(lib.dctr_embed_fwd(cplan, P(X), X.stride(0), B, P(out0), plan.ld_out, P(wide0), 1, P(fm0), None,
                               plan.units_ptr(), len(plan.units), None, None, P(fms0), 16, s))
    yp0, gl0, loss0 = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty((), **f32)
    gx0 = torch.empty(B, plan.ld_out, **f32)
    L.check(lib.dctr_mlp_train_step(ctypes.byref(b.desc), P(out0), plan.ld_out, B, P(wide0), P(fm0), P(m.out.bias), P(y),
                                    P(yp0), P(loss0), P(gl0), None, P(gx0), plan.ld_out, P(b.ws), 1, None, s))
    torch.cuda.synchronize()
    h0 = [h.clone() for h in b.hs]
    dh0 = [h.clone() for h in b.dhs]
    out1, fms1 = torch.zeros(B, plan.ld_out, **f32), torch.empty(B, 16, **f32)
    yp1, gl1, gx1 = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, plan.ld_out, **f32)
    L.check(lib.dctr_embed_tower_train_step(cplan, P(X), X.stride(0), ctypes.byref(b.desc), B, 1, P(m.out.bias), P(y),
                                            P(yp1), P(gl1), P(gx1), plan.ld_out, P(out1), plan.ld_out, P(fms1), 16,
                                            P(plan.err_flag(X.device)), P(b.ws), s))
    torch.cuda.synchronize()
    W = plan.width
    assert torch.equal(out1[:, :W], out0[:, :W]), "gathered rows"
    assert float(out1[:, W:].abs().max()) == 0.0, "padding columns of `out` must be written as zeros"
    assert torch.equal(fms1, fms0), "sum_f e"
    assert torch.equal(yp1, yp0), "y_pred (max %.3e)" % float((yp1 - yp0).abs().max())
    assert torch.equal(gl1, gl0), "g_logit"
    assert torch.equal(gx1[:, :W], gx0[:, :W]), "gx"
    for a, c in zip(b.hs, h0):
        assert torch.equal(a, c)
    for a, c in zip(b.dhs, dh0):
        assert torch.equal(a, c)


def test_step_signal_releases_a_waiter():
    """dctr_step_signal (the signalling half as a launch of its own) against dctr_step_wait: the n-th wait returns when the
    n-th signal has been given; no time-out bit is raised."""
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    sync = torch.zeros(L.SYNC_INTS, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    torch.cuda.synchronize()
    for _ in range(3):
        L.check(lib.dctr_step_wait(P(sync), L.SYNC_UPDATE, 20000, L.stream_handle(sync.device)))
        with torch.cuda.stream(side):
            L.check(lib.dctr_step_signal(P(sync), L.SYNC_UPDATE, ctypes.c_void_p(side.cuda_stream)))
    torch.cuda.synchronize()
    v = sync.cpu()
    assert int(v[L.SYNC_ERR]) == 0
    assert int(v[4 * L.SYNC_UPDATE]) == 3 and int(v[4 * L.SYNC_UPDATE + 1]) == 3
