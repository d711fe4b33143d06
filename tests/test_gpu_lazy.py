"""GPU: the exact lazy regularised / Adam table update (csrc/lazy.hip) against trajectories of the REAL reference.

The fixtures (oracle/make_golden.py, ``lazy_*``) hold 8 training steps of the reference with its default kind of
configuration -- an L2 term on every table (l2 = 1e-3 here) under SGD / Adagrad / Adam, and Adam without L2 -- on
small batches over small vocabularies: most rows are untouched by a step and touched again a few steps later.  The
reference updates every row at every step (dense gradients, O(vocabulary)); the drop-in replays each row's steps when
it is next needed (O(batch)) and must land on the same parameters, losses, predictions and optimizer state.
Tolerance: 2e-5 x max|reference| (fp32 re-association; the recurrences are the same)."""
import os

import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = [("lazy_deepfm", t) for t in ("sgd", "adagrad", "adam", "adam0")] + \
        [("lazy_dcn", t) for t in ("sgd", "adagrad", "adam", "adam0")]


def _close(tag, got, ref, tol=2e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
    err = max_abs(got, ref)
    assert err <= tol * scale, "%s: max|d| = %.3e (scale %.3e)" % (tag, err, scale)


def _run(name, tag, steps=None):
    g = load_golden(name)
    ex = g["extra"]
    l2 = 0.0 if tag == "adam0" else 1e-3
    m = build_model(g["spec"], DEV, l2=l2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.compile("adam" if tag == "adam0" else tag, "binary_crossentropy", metrics=[])
    m.train()
    plan = m.model_plan()
    assert plan.update == ("lazy", "adam" if tag == "adam0" else tag), plan.update
    bce, tot = [], []
    Xs, ys = ex["lazy_X"], ex["lazy_y"]
    for i in range(len(Xs) if steps is None else steps):
        loss, total, _ = m._train_step(torch.from_numpy(Xs[i]).to(DEV), torch.from_numpy(ys[i]).to(DEV))
        bce.append(float(loss.item()))
        tot.append(float(total.item()))
    return g, m, bce, tot


@pytest.mark.parametrize("name,tag", CASES)
def test_lazy_trajectory_matches_reference(name, tag):
    g, m, bce, tot = _run(name, tag)
    ex = g["extra"]
    np.testing.assert_allclose(bce, ex["lazy_%s_bce" % tag], rtol=2e-5)
    np.testing.assert_allclose(tot, ex["lazy_%s_total" % tag], rtol=2e-5)
    sd = m.state_dict()                     # flushes: every row replayed to the current step
    for k, v in ex.items():
        if k.startswith("lazy_%s/" % tag):
            _close(k, sd[k[len("lazy_%s/" % tag):]].cpu().numpy(), v)
    # predict() on the flushed tables
    m.eval()
    with torch.no_grad():
        pred = m(torch.from_numpy(ex["lazy_X"][0]).to(DEV))
    _close("pred", pred.cpu().numpy().reshape(-1, 1), ex["lazy_%s_pred" % tag])
    # the torch optimizer's state tensors ARE the kernels' state: optimizer.state_dict() stays meaningful
    p0 = m.embedding_dict[g["spec"]["dnn_columns"][0]["embedding_name"]].weight
    st = m.optim.state[p0]
    for key in ("sum", "exp_avg", "exp_avg_sq"):
        ref = ex.get("lazy_%s_state_%s" % (tag, key))
        if ref is not None:
            _close("state." + key, st[key].cpu().numpy(), ref)
    if tag.startswith("adam"):
        assert float(st["step"]) == len(ex["lazy_X"])


def test_lazy_mid_run_predict_and_resume():
    """A flush in the middle of training (predict between steps) must not disturb the trajectory."""
    name, tag = "lazy_deepfm", "adam"
    g = load_golden(name)
    ex = g["extra"]
    m = build_model(g["spec"], DEV, l2=1e-3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
    m.compile(tag, "binary_crossentropy", metrics=[])
    Xs, ys = ex["lazy_X"], ex["lazy_y"]
    for i in range(len(Xs)):
        m.train()
        m._train_step(torch.from_numpy(Xs[i]).to(DEV), torch.from_numpy(ys[i]).to(DEV))
        if i in (2, 5):
            m.eval()
            with torch.no_grad():
                m(torch.from_numpy(Xs[0]).to(DEV))
    sd = m.state_dict()
    for k, v in ex.items():
        if k.startswith("lazy_adam/"):
            _close(k, sd[k[len("lazy_adam/"):]].cpu().numpy(), v)


def test_lazy_equals_dense_path(monkeypatch):
    """Same model, same steps, DCTR_LAZY_UPDATE=0 (exact dense gradients + torch.optim): same parameters."""
    g, m, bce, tot = _run("lazy_deepfm", "adagrad")
    monkeypatch.setenv("DCTR_LAZY_UPDATE", "0")
    g2 = load_golden("lazy_deepfm")
    m2 = build_model(g2["spec"], DEV, l2=1e-3)
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in g2["params"].items()})
    m2.compile("adagrad", "binary_crossentropy", metrics=[])
    m2.train()
    assert m2.model_plan().update == ("dense",)
    for i in range(len(g2["extra"]["lazy_X"])):
        m2._train_step(torch.from_numpy(g2["extra"]["lazy_X"][i]).to(DEV), torch.from_numpy(g2["extra"]["lazy_y"][i]).to(DEV))
    a, b = m.state_dict(), m2.state_dict()
    for k in a:
        _close(k, a[k].cpu().numpy(), b[k].cpu().numpy())


def test_default_kwargs_fit_runs_lazy():
    """Reference defaults (l2_reg_embedding = l2_reg_linear = 1e-5, 'adam') through fit / predict."""
    g = load_golden("lazy_deepfm")
    from helpers import feature_columns
    from deepctr_torch.models import DeepFM
    cols = feature_columns(g["spec"]["dnn_columns"])
    m = DeepFM(cols, cols, dnn_hidden_units=(16, 8), device=DEV)        # default l2 = 1e-5
    m.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy"])
    X = np.concatenate(list(g["extra"]["lazy_X"]), 0)
    y = np.concatenate(list(g["extra"]["lazy_y"]), 0)
    x = {c["name"]: X[:, i] for i, c in enumerate(g["spec"]["dnn_columns"])}
    hist = m.fit(x, y, batch_size=32, epochs=2, verbose=0, validation_split=0.2)
    assert m.model_plan().update == ("lazy", "adam")
    assert len(hist.history["loss"]) == 2 and np.isfinite(hist.history["loss"]).all()
    pred = m.predict(x, batch_size=64)
    assert pred.shape == (X.shape[0], 1) and np.all((pred > 0) & (pred < 1))


@pytest.mark.parametrize("opt", ["adam", "adagrad", "rmsprop"])
def test_lazy_long_gaps_equal_dense_path(opt, monkeypatch):
    """200 steps of batch 8 over 3000-row vocabularies: most rows wait tens to hundreds of steps between two touches
    (the steady state of a big table).  The replayed trajectories must equal the exact dense path's."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    gen = torch.Generator().manual_seed(3)
    V, B, steps = 3000, 8, 200
    cols = [SparseFeat("a", V, 8), SparseFeat("b", V + 7, 8), DenseFeat("d", 2)]
    X = torch.cat([torch.randint(0, V, (steps * B, 2), generator=gen).float(), torch.rand(steps * B, 2, generator=gen)], 1)
    # a few hot ids so that some rows ARE touched often
    X[::3, 0] = 5.0
    y = torch.randint(0, 2, (steps * B,), generator=gen).float()
    X, y = X.to(DEV), y.to(DEV)
    finals = []
    for lazy in ("1", "0"):
        monkeypatch.setenv("DCTR_LAZY_UPDATE", lazy)
        m = DeepFM(cols, cols, dnn_hidden_units=(8,), l2_reg_embedding=1e-3, l2_reg_linear=1e-3, init_std=0.1, seed=7,
                   device=DEV)
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        assert m.model_plan().update[0] == ("lazy" if lazy == "1" else "dense")
        for i in range(steps):
            m._train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
        finals.append({k: v.clone() for k, v in m.state_dict().items()})
    # (RMSprop divides by the root of a decaying average that is tiny for rarely touched rows: its normalised step turns
    # the last-bit differences between the two routes' summation orders into the largest differences of the three -- 1.1e-4
    # on weights of scale 1 once the gather stopped contracting multiply-adds (round 4); Adam / Adagrad stay below 1e-4)
    tol = 3e-4 if opt == "rmsprop" else 1e-4
    for k in finals[0]:
        _close(k, finals[0][k].cpu().numpy(), finals[1][k].cpu().numpy(), tol=tol)


@pytest.mark.parametrize("opt,sweep_k", [("adam", "256"), ("adam", "7"), ("adam", "0"), ("adagrad", "7"), ("sgd", "5")])
def test_ordered_catchup_and_sweep_equal_dense_path(opt, sweep_k, monkeypatch):
    """Batches of 256 over 20 000-row vocabularies (mean gap ~80 steps): large enough that the catch-up deals its entries by
    gap (k_lazy_order, B >= 64) and that the whole wave walks the steps together (replay_in_step); the per-step sweep with
    the default K, with a K the run wraps around many times, and without.  Every (row, step) must be applied exactly once,
    in order, whoever applies it -- sweep, catch-up or the final flush: the exact dense path's parameters."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    gen = torch.Generator().manual_seed(5)
    V, B, steps = 20000, 256, 60
    cols = [SparseFeat("a", V, 16), SparseFeat("b", V // 3, 16), SparseFeat("c", 50, 16), DenseFeat("d", 2)]
    X = torch.cat([torch.randint(0, V, (steps * B, 1), generator=gen).float(),
                   torch.randint(0, V // 3, (steps * B, 1), generator=gen).float(),
                   torch.randint(0, 50, (steps * B, 1), generator=gen).float(), torch.rand(steps * B, 2, generator=gen)], 1)
    y = torch.randint(0, 2, (steps * B,), generator=gen).float()
    X, y = X.to(DEV), y.to(DEV)
    finals = []
    for lazy in ("1", "0"):
        monkeypatch.setenv("DCTR_LAZY_UPDATE", lazy)
        monkeypatch.setenv("DCTR_LAZY_SWEEP_K", sweep_k)
        m = DeepFM(cols, cols, dnn_hidden_units=(16,), l2_reg_embedding=1e-3, l2_reg_linear=1e-3, init_std=0.1, seed=7,
                   device=DEV)
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        assert m.model_plan().update[0] == ("lazy" if lazy == "1" else "dense")
        for i in range(steps):
            m._train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
            if lazy == "1" and i == steps // 2:
                m.state_dict()                      # a flush in the middle: the sweep goes on from wherever the counter is
        if lazy == "1":
            lz = m.model_plan().lazy
            assert lz.sweep_k == int(sweep_k)
            if int(sweep_k) > 0 and int(sweep_k) <= steps:      # every window came by: no row is more than K steps behind
                t = int(lz.step.item())
                assert all(int((t - st).max().item()) <= int(sweep_k) for st in lz.stamps)
        finals.append({k: v.clone() for k, v in m.state_dict().items()})
    for k in finals[0]:
        _close(k, finals[0][k].cpu().numpy(), finals[1][k].cpu().numpy(), tol=1e-4)


@pytest.mark.parametrize("name,opt", [("lazy_deepfm", "adam"), ("lazy_deepfm", "adagrad"), ("lazy_dcn", "adagrad")])
def test_fit_graph_replays_leave_tables_flushed(monkeypatch, name, opt):
    """Every batch full-size (sample_num % batch_size == 0) over 3 epochs: after the first epoch all train steps are
    hipGraph replays, which never pass through LazyState.apply().  The epoch-end flush / state_dict() / predict() must
    still bring every row up to date (round-1 advisor finding: the host-side dirty flag stayed False) -- same tables,
    optimizer state and predictions as the eager fit."""
    g = load_golden(name)
    Xs, ys = g["extra"]["lazy_X"], g["extra"]["lazy_y"]
    X, y = np.concatenate(list(Xs), 0), np.concatenate(list(ys), 0)
    bs = Xs[0].shape[0]
    assert X.shape[0] % bs == 0
    names = []
    for c in g["spec"]["linear_columns"] + g["spec"]["dnn_columns"]:
        if c["name"] not in names:
            names.append(c["name"])
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("DCTR_FIT_GRAPH", flag)
        m = build_model(g["spec"], DEV, l2=1e-3)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile(opt, "binary_crossentropy", metrics=[])
        fi = m.feature_index
        x = {nm: X[:, fi[nm][0]] for nm in names}
        hist = m.fit(x, y, batch_size=bs, epochs=3, verbose=0, shuffle=False)
        used = m._fit_graph is not None and m._fit_graph.get("graph") is not None
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        pred = m.predict(x, batch_size=bs)
        runs.append((sd, dict(hist.history), pred, used))
    (a, ha, pa, ua), (b, hb, pb, ub) = runs
    assert ua and not ub
    for k in a:
        _close(k, a[k].cpu().numpy(), b[k].cpu().numpy(), tol=1e-6)
    np.testing.assert_allclose(ha["loss"], hb["loss"], rtol=1e-6)
    assert max_abs(pa, pb) <= 1e-6


def test_side_streams_never_alias_after_many_stream_objects():
    """torch hands out its 32 pool streams round-robin: a process that has created a few dozen stream objects (a dozen
    models used to do that) gets the SAME queue again under a new object.  The package's side streams -- pre-pass,
    weight-gradient fork, batch staging, warm-up, capture -- must stay pairwise distinct and distinct from whatever
    torch hands out next, or two 'different' streams of one hipGraph capture are one queue (a segmentation fault at the
    first replay, once 14 earlier tests of this file had run)."""
    from deepctr_torch._hip import streams
    junk = [torch.cuda.Stream() for _ in range(45)]           # wrap the pool around
    got = [streams.side_stream(DEV, r) for r in streams.ROLES]
    assert len(set(s.stream_id for s in got)) == len(streams.ROLES)
    assert all(streams.side_stream(DEV, r) is s for r, s in zip(streams.ROLES, got))      # created once
    # and a graphed fit still replays correctly with the pool wrapped around
    g = load_golden("lazy_deepfm")
    Xs, ys = g["extra"]["lazy_X"], g["extra"]["lazy_y"]
    X, y = np.concatenate(list(Xs), 0), np.concatenate(list(ys), 0)
    bs = Xs[0].shape[0]
    names = []
    for c in g["spec"]["linear_columns"] + g["spec"]["dnn_columns"]:
        if c["name"] not in names:
            names.append(c["name"])
    hists = []
    for flag in ("1", "0"):
        os.environ["DCTR_FIT_GRAPH"] = flag
        try:
            m = build_model(g["spec"], DEV, l2=1e-3)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
            m.compile("adagrad", "binary_crossentropy", metrics=[])
            fi = m.feature_index
            hists.append(m.fit({nm: X[:, fi[nm][0]] for nm in names}, y, batch_size=bs, epochs=3, verbose=0,
                               shuffle=False).history["loss"])
        finally:
            os.environ.pop("DCTR_FIT_GRAPH", None)
    np.testing.assert_allclose(hists[0], hists[1], rtol=1e-6)
    del junk


@pytest.mark.parametrize("opt", ["adam", "adagrad", "sgd"])
@pytest.mark.parametrize("ids", ["uniform", "hot"])
def test_step_inside_the_sorted_update_equals_the_two_pass_route(monkeypatch, opt, ids):
    """Round 6: the data-gradient step of the lazily regularised / Adam tables runs at the row inside the sorted update
    (dctr_embed_update_lazy, csrc/update_kernels.hpp DCTR_UPD_LAZY) instead of dctr_embed_update(ACCUM) + dctr_lazy_apply.
    Same sums in the same order, the same optimizer arithmetic (csrc/lazy_opt.hpp): after 6 steps at batch 4096 the
    tables, both optimizer moments and the losses of the two routes agree to a few ulps -- also with a HOT id (85 % of a
    column's entries: the partition overflows the pre-pass's bucket and takes the update kernel's general path; its ~3 500
    gradient strips are then summed in another grouping than the accumulate pass uses -- sqrt(n) ulps of reordering noise in
    the first moment of that one row, hence 4e-6 there instead of 1e-6)."""
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    B, V, F, D = 4096, 20000, 6, 16
    gen = torch.Generator().manual_seed(3)
    X_ids = torch.randint(0, V, (8 * B, F), generator=gen)
    if ids == "hot":
        hot = torch.rand(8 * B, F, generator=gen) < 0.85
        X_ids = torch.where(hot, torch.full_like(X_ids, 7), X_ids)
    X = torch.cat([X_ids.float(), torch.rand(8 * B, 3, generator=gen)], 1).to(DEV)
    y = torch.randint(0, 2, (8 * B,), generator=gen).float().to(DEV)
    runs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("DCTR_LAZY_FUSED_APPLY", fused)
        cols = [SparseFeat("C%d" % i, V, D) for i in range(F)] + [DenseFeat("I%d" % i, 1) for i in range(3)]
        m = DeepFM(cols, cols, dnn_hidden_units=(64, 32), l2_reg_linear=1e-4, l2_reg_embedding=1e-4, init_std=0.05, seed=5,
                   device=DEV)
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        assert m.model_plan().update[0] == "lazy"
        calls = []
        if fused == "1":
            from deepctr_torch._hip import lib as L
            real = L.lib().dctr_embed_update_lazy
            monkeypatch.setattr(L.lib(), "dctr_embed_update_lazy", lambda *a: (calls.append(1), real(*a))[1], raising=False)
        losses = [m._train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])[0].item() for i in range(6)]
        if fused == "1":
            assert len(calls) == 6, "the fused route was not taken"
            monkeypatch.undo()
        torch.cuda.synchronize()
        m.model_plan().check_ids()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}      # (flushes every row to the current step)
        st = {}
        for k, p in m.named_parameters():
            for key, v in m.optim.state.get(p, {}).items():
                if torch.is_tensor(v) and v.numel() > 1:
                    st[k + "/" + key] = v.detach().cpu().clone()
        runs.append((losses, sd, st))
    (la, sa, ta), (lb, sb, tb) = runs
    bar = 4e-6 if ids == "hot" else 1e-6
    np.testing.assert_allclose(la, lb, rtol=bar)
    for k in sb:
        err = float((sa[k] - sb[k]).abs().max())
        assert err <= bar * max(1.0, float(sb[k].abs().max())), "%s: %.3e" % (k, err)
    for k in tb:
        err = float((ta[k] - tb[k]).abs().max())
        assert err <= bar * max(1.0, float(tb[k].abs().max())) + 1e-12, "%s: %.3e" % (k, err)


def _dp_lazy_worker(rank, port):
    """(child process: RCCL is initialised and torn down here, not in the pytest process -- a hipGraph replay in a process that
    had a process group destroyed earlier segfaulted inside the runtime)"""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(ROOT, "deepctr-torch_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import torch.distributed as dist
    from deepctr_torch import parallel as par
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    DEV = "cuda:0"
    B, V, F, D = 2048, 5000, 6, 16
    gen = torch.Generator().manual_seed(9)
    X = torch.cat([torch.randint(0, V, (6 * B, F), generator=gen).float(), torch.rand(6 * B, 3, generator=gen)], 1).to(DEV)
    y = torch.randint(0, 2, (6 * B,), generator=gen).float().to(DEV)

    def build():
        cols = [SparseFeat("C%d" % i, V, D) for i in range(F)] + [DenseFeat("I%d" % i, 1) for i in range(3)]
        m = DeepFM(cols, cols, dnn_hidden_units=(64, 32), l2_reg_linear=1e-4, l2_reg_embedding=1e-4, init_std=0.05, seed=5, device=DEV)
        m.compile("adam", "binary_crossentropy", metrics=[])
        m.train()
        assert m.model_plan().update[0] == "lazy"
        return m

    ref = build()
    os.environ["DCTR_FUSED_STEP"] = "0"       # (the trainer's own route for the dense parameters: autograd + torch.optim)
    ref_losses = [float(ref._train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])[0].item()) for i in range(6)]
    os.environ.pop("DCTR_FUSED_STEP")
    ref_sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        m = build()
        tr = par.DataParallelTrainer(m)
        assert tr._lazy and m.model_plan().update[0] == "lazy"
        losses = [float(tr.train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])[0].item()) for i in range(6)]
        tr.close()
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    finally:
        dist.destroy_process_group()
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    for k, v in ref_sd.items():
        err = float((sd[k] - v).abs().max())
        assert err <= 2e-5 * max(1.0, float(v.abs().max())), "%s: %.3e" % (k, err)


def test_data_parallel_trainer_keeps_the_lazy_update():
    """Round 6: the reference's default kwargs (L2 on the tables, adam) under the replicated-tables trainer stay on the lazy
    update -- a replica catches up its own batch's rows before its gather and the global batch's rows before the data-gradient
    step.  One rank (RCCL at world size 1, in a child process) against the single-process train step on the same batches: same
    losses, tables and moments (the trainer's step goes through torch.optim for the dense parameters like the reference step it
    is compared with: 2e-5)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_lazy_worker, args=(port,), nprocs=1, join=True)
