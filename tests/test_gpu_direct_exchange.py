"""The direct exchange of the table-sharded step (deepctr_torch.parallel.DirectExchange: copies into the peers' IPC-mapped
receive buffers + arrival words, no host-issued collective, the whole step one hipGraph) with N > 1 RANKS FOR REAL: N
processes that share the one GPU of this box -- RCCL refuses two ranks on one device, a hand-written exchange does not.
Every rank runs the real kernels on its own shard of the tables and its own slice of the global batch; after a few steps
and ``gather_tables()`` every rank must hold the parameters ONE process reaches on the concatenated batch with the fused
single-GPU step (itself pinned to the reference: tests/test_gpu_deepfm.py, test_gpu_full_golden.py).
gloo carries the IPC handshake and the final table broadcast only."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F_, V_, D_, ND_, B_ = 7, 5000, 16, 3, 256
STEPS = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch(step, world):
    g = torch.Generator().manual_seed(100 + step)
    ids = torch.randint(0, V_, (world * B_, F_), generator=g).float()
    X = torch.cat([ids, torch.rand(world * B_, ND_, generator=g)], 1)
    y = torch.randint(0, 2, (world * B_,), generator=g).float()
    return X, y


def _model(dev):
    for p in (os.path.join(ROOT, "deepctr-torch_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("C%d" % i, V_, D_) for i in range(F_)] + [DenseFeat("I%d" % i, 1) for i in range(ND_)]
    return DeepFM(cols, cols, dnn_hidden_units=(64, 32), l2_reg_linear=0, l2_reg_embedding=0, init_std=0.05, seed=7,
                  device=dev)


def _worker(rank, world, port, opt_name, graphs, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = "cuda:0"
        m = _model(dev)
        from deepctr_torch.parallel import ShardedTrainer
        m.compile(opt_name, "binary_crossentropy", metrics=[])
        m.train()
        tr = ShardedTrainer(m, exchange="direct", use_graphs=False)
        batches = [_batch(step, world) for step in range(STEPS)]
        mine = [(Xg[rank * B_:(rank + 1) * B_].contiguous().to(dev), yg[rank * B_:(rank + 1) * B_].contiguous().to(dev))
                for Xg, yg in batches]
        losses = []
        for step in range(STEPS):
            if graphs and step == 2:
                tr.set_use_graphs(True)          # eager steps first, then the captured whole-step graph
            nxt = mine[step + 1][0] if step + 1 < STEPS and step != 1 else None      # (step 1 -> 2 is NOT announced)
            losses.append(tr.train_step(mine[step][0], mine[step][1], next_xb=nxt)[0].clone())   # (a replay's outputs are static)
        torch.cuda.synchronize()
        tr.gather_tables()
        tr.close()
        m.model_plan().check_ids()
        torch.save({"sd": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    "loss": torch.stack([l.reshape(()) for l in losses]).cpu()}, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("opt_name", ["adagrad", "sgd"])
def test_direct_exchange_ranks_on_one_gpu_equal_the_single_process_step(tmp_path, opt_name, world, graphs):
    if opt_name == "sgd" and (world == 3 or not graphs):
        pytest.skip("one SGD case is enough")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, opt_name, graphs, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for r in range(1, world):
        for k in ranks[0]["sd"]:
            assert torch.equal(ranks[0]["sd"][k], ranks[r]["sd"][k]), "replicas differ after gather_tables: %s" % k
    # the single-process fused step on the concatenated batch
    dev = "cuda:0"
    ref = _model(dev)
    ref.compile(opt_name, "binary_crossentropy", metrics=[])
    ref.train()
    ref_loss = []
    for step in range(STEPS):
        Xg, yg = _batch(step, world)
        ref_loss.append(ref._train_step(Xg.to(dev), yg.to(dev))[0].reshape(()))
    torch.cuda.synchronize()
    ref_sd = {k: v.detach().cpu() for k, v in ref.state_dict().items()}
    got_loss = sum(r["loss"] for r in ranks)
    assert torch.allclose(got_loss, torch.stack(ref_loss).cpu(), rtol=2e-5), (got_loss, ref_loss)
    for k, v in ref_sd.items():
        err = float((ranks[0]["sd"][k] - v).abs().max())
        assert err <= 2e-6 * max(1.0, float(v.abs().max())) + 2e-7, "%s: %.3e" % (k, err)


# ---- S steps per hipGraph: ShardedTrainer.train_block ------------------------------------------------------------------
BLK_S, BLK_N = 2, 3            # three blocks of two steps behind two single steps: eager block, captured block, replay


def _worker_block(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = "cuda:0"
        m = _model(dev)
        from deepctr_torch.parallel import ShardedTrainer
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        tr = ShardedTrainer(m, exchange="direct", use_graphs=False)
        n = 2 + BLK_S * BLK_N
        batches = [_batch(step, world) for step in range(n)]
        Xr = torch.stack([Xg[rank * B_:(rank + 1) * B_] for Xg, _ in batches]).to(dev)      # [n, B, C] resident
        yr = torch.stack([yg[rank * B_:(rank + 1) * B_] for _, yg in batches]).to(dev)
        losses = []
        for step in range(2):
            losses.append(tr.train_step(Xr[step], yr[step], next_xb=Xr[step + 1])[0].clone())
        tr.set_use_graphs(True)
        for b in range(BLK_N):
            lo = 2 + b * BLK_S
            nxt = Xr[lo + BLK_S] if lo + BLK_S < n else None
            losses.append(tr.train_block(Xr[lo:lo + BLK_S], yr[lo:lo + BLK_S], next_first=nxt)[0].clone())
        torch.cuda.synchronize()
        tr._dx.check()
        tr.gather_tables()
        tr.close()
        m.model_plan().check_ids()
        torch.save({"sd": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    "loss": torch.stack([l.reshape(()) for l in losses]).cpu()}, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_train_block_equals_the_single_process_steps(tmp_path):
    """Two ranks sharing the GPU, the steps behind the second one as blocks of BLK_S steps per hipGraph (first block eager,
    second captured and replayed, third a pure replay): parameters and the blocks' last losses equal the single-process
    fused step on the concatenated batches."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_block, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for k in ranks[0]["sd"]:
        assert torch.equal(ranks[0]["sd"][k], ranks[1]["sd"][k]), "replicas differ after gather_tables: %s" % k
    dev = "cuda:0"
    ref = _model(dev)
    ref.compile("adagrad", "binary_crossentropy", metrics=[])
    ref.train()
    n = 2 + BLK_S * BLK_N
    ref_loss = []
    for step in range(n):
        Xg, yg = _batch(step, world)
        ref_loss.append(ref._train_step(Xg.to(dev), yg.to(dev))[0].reshape(()))
    torch.cuda.synchronize()
    want = torch.stack([ref_loss[0], ref_loss[1]] + [ref_loss[2 + (b + 1) * BLK_S - 1] for b in range(BLK_N)]).cpu()
    got = sum(r["loss"] for r in ranks)
    assert torch.allclose(got, want, rtol=2e-5), (got, want)
    for k, v in ref.state_dict().items():
        v = v.detach().cpu()
        err = float((ranks[0]["sd"][k] - v).abs().max())
        assert err <= 2e-6 * max(1.0, float(v.abs().max())) + 2e-7, "%s: %.3e" % (k, err)


# ---- the exchange's self-test (what fit() under torchrun and bench.py run before they trust it at N > 1) -------------------
def _worker_selftest(rank, world, port, sabotage, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        for p in (os.path.join(ROOT, "deepctr-torch_amd"),):
            if p not in sys.path:
                sys.path.insert(0, p)
        from deepctr_torch import parallel as par
        if sabotage and rank == 1:
            # one rank sees wrong bytes (here: it compares against a pattern nobody wrote): EVERY rank must fall back
            real = par.DirectExchange.self_test
            par.DirectExchange.self_test = lambda self, rounds=3: real(self, rounds) + 7
        got = par.resolve_exchange("auto", "cuda:0", verbose=False)
        dx = par.DirectExchange(None, world, rank, torch.device("cuda:0"), 32, 16, 2, 64,
                                dense_src=torch.zeros(64, device="cuda:0"))
        n_bad = dx.self_test(rounds=4)
        torch.save({"got": got, "n_bad": n_bad}, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sabotage", [False, True], ids=["clean", "one_rank_fails"])
def test_exchange_self_test_and_resolution(tmp_path, sabotage):
    """parallel.resolve_exchange('auto') with three ranks on the GPU: the direct exchange when DirectExchange.self_test sees
    every byte right on every rank, RCCL on EVERY rank as soon as one rank reports a mismatch; the self-test itself (kernel
    pushes into the peers' buffers, pulls through the pointer table, four rounds over the same addresses) counts zero wrong
    elements."""
    world = 3
    port = _free_port()
    mp.spawn(_worker_selftest, args=(world, port, sabotage, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for r in ranks:
        assert r["got"][0] == ("rccl" if sabotage else "direct"), r["got"]
        assert r["n_bad"] == (7 if (sabotage and r is ranks[1]) else 0)


# ---- fit() itself through the direct exchange: groups of steps as one hipGraph each ----------------------------------------
FIT_B, FIT_N, FIT_S = 64, 64 * 2 * 11 + 37, 3     # per-rank batch; 11 global batches of 128 and a ragged tail of 37 rows


def _fit_data():
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V_, (FIT_N, F_), generator=g).float()
    X = torch.cat([ids, torch.rand(FIT_N, ND_, generator=g)], 1)
    y = torch.randint(0, 2, (FIT_N,), generator=g).float()
    return X, y


def _worker_fit(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", DCTR_FIT_STEPS_PER_GRAPH=str(FIT_S))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # (RCCL refuses two ranks on one device)
    try:
        torch.cuda.set_device(0)
        m = _model("cuda:0")
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        X, y = _fit_data()
        torch.manual_seed(77)
        hist = m.fit(X.to("cuda:0"), y.to("cuda:0"), batch_size=FIT_B, epochs=2, verbose=0, shuffle=True)
        tr = m._dist_trainer
        torch.save({"sd": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, "loss": list(hist.history["loss"]),
                    "trainer": type(tr).__name__, "exchange": tr.exchange, "note": tr.tr.exchange_note,
                    "blocks": tr.blocks(torch.device("cuda:0")), "had_block": getattr(tr.tr, "last_block_loss", None) is not None},
                   os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_fit_through_the_direct_exchange_equals_single_process_fit(tmp_path):
    """``model.fit()`` with two ranks on the GPU: the exchange resolves to 'direct' through its self-test, the epoch runs two
    eager steps, then blocks of three steps per hipGraph (ShardedTrainer.train_block), the left-over full batches one by one
    and the ragged tail on gathered tables -- History and parameters of ONE process fitting the global batch size."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_fit, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for r in ranks:
        assert r["trainer"] == "_Sharded" and r["exchange"] == "direct" and "self-test passed" in r["note"], r
        assert r["blocks"] == FIT_S and r["had_block"]
    ref = _model("cuda:0")
    ref.compile("adagrad", "binary_crossentropy", metrics=[])
    X, y = _fit_data()
    torch.manual_seed(77)
    os.environ["DCTR_FIT_DISTRIBUTED"] = "0"
    try:
        hist = ref.fit(X.to("cuda:0"), y.to("cuda:0"), batch_size=FIT_B * world, epochs=2, verbose=0, shuffle=True)
    finally:
        os.environ.pop("DCTR_FIT_DISTRIBUTED", None)
    import numpy as np
    for r in ranks:
        np.testing.assert_allclose(r["loss"], hist.history["loss"], rtol=2e-5)
    for k, v in ref.state_dict().items():
        v = v.detach().cpu()
        err = float((ranks[0]["sd"][k] - v).abs().max())
        assert err <= 2e-6 * max(1.0, float(v.abs().max())) + 2e-7, "%s: %.3e" % (k, err)
        assert torch.equal(ranks[0]["sd"][k], ranks[1]["sd"][k]), "replicas differ: %s" % k


# ---- general table groups: pooled VarLen fields and tables shared through embedding_name, sharded (round 6) ----------------
def _pooled_cols():
    for p in (os.path.join(ROOT, "deepctr-torch_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    cols = [SparseFeat("C%d" % i, V_, D_) for i in range(4)]
    cols.append(SparseFeat("C4", V_, D_, embedding_name="C0"))                       # shares C0's table
    cols += [DenseFeat("I%d" % i, 1) for i in range(2)]
    cols.append(VarLenSparseFeat(SparseFeat("hist", V_, D_, embedding_name="C1"), maxlen=4, combiner="mean"))   # over C1's table
    cols.append(VarLenSparseFeat(SparseFeat("tags", 300, D_), maxlen=3, combiner="sum"))
    cols.append(VarLenSparseFeat(SparseFeat("kw", 200, D_), maxlen=5, combiner="max", length_name="kw_len"))
    return cols


def _pooled_model(dev):
    from deepctr_torch.models import DeepFM
    cols = _pooled_cols()
    return DeepFM(cols, cols, dnn_hidden_units=(64, 32), l2_reg_linear=0, l2_reg_embedding=0, init_std=0.05, seed=7, device=dev)


def _pooled_batch(step, world):
    g = torch.Generator().manual_seed(300 + step)
    n = world * B_
    ids = torch.randint(0, V_, (n, 5), generator=g).float()
    dense = torch.rand(n, 2, generator=g)
    hist = torch.randint(1, V_, (n, 4), generator=g) * (torch.arange(4)[None, :] < torch.randint(0, 5, (n, 1), generator=g))
    tags = torch.randint(1, 300, (n, 3), generator=g) * (torch.arange(3)[None, :] < torch.randint(0, 4, (n, 1), generator=g))
    kw = torch.randint(0, 200, (n, 5), generator=g)
    kw_len = torch.randint(1, 6, (n, 1), generator=g)
    # (inputs.py:99-123: columns in feature order; a VarLen's positions, then -- behind the first VarLen that names it -- its
    # length column)
    X = torch.cat([ids, dense, hist.float(), tags.float(), kw.float(), kw_len.float()], 1)
    y = torch.randint(0, 2, (n,), generator=g).float()
    return X, y


def _worker_pooled(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = "cuda:0"
        m = _pooled_model(dev)
        from deepctr_torch.parallel import ShardedTrainer
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        tr = ShardedTrainer(m, exchange="direct", use_graphs=False)
        lay = tr.layout
        assert not lay.simple and lay.n_ids > lay.n_slots
        batches = [_pooled_batch(step, world) for step in range(STEPS)]
        mine = [(Xg[rank * B_:(rank + 1) * B_].contiguous().to(dev), yg[rank * B_:(rank + 1) * B_].contiguous().to(dev))
                for Xg, yg in batches]
        losses = []
        for step in range(STEPS):
            if step == 2:
                tr.set_use_graphs(True)
            nxt = mine[step + 1][0] if step + 1 < STEPS else None
            losses.append(tr.train_step(mine[step][0], mine[step][1], next_xb=nxt)[0].clone())
        torch.cuda.synchronize()
        tr.gather_tables()
        tr.close()
        m.model_plan().check_ids()
        torch.save({"sd": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    "loss": torch.stack([l.reshape(()) for l in losses]).cpu(),
                    "owners": list(lay.group_owner), "n_slots": lay.n_slots, "n_ids": lay.n_ids},
                   os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pooled_and_shared_tables_are_sharded_too(tmp_path, world):
    """DeepFM with a table shared through embedding_name, a mean history over another field's table, a sum-pooled and a
    max-pooled VarLen field (length column): ShardedTrainer owns table GROUPS, the owner pools locally and ships one row per
    field, the owner's update is the deterministic general one -- parameters and losses of the single-process step on the
    concatenated batch (whose pooled lookup / update are pinned to the reference: tests/test_gpu_update_general.py)."""
    port = _free_port()
    mp.spawn(_worker_pooled, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    assert len(set(ranks[0]["owners"])) == min(world, len(ranks[0]["owners"]))       # every rank owns a group
    for r in range(1, world):
        for k in ranks[0]["sd"]:
            assert torch.equal(ranks[0]["sd"][k], ranks[r]["sd"][k]), "replicas differ after gather_tables: %s" % k
    dev = "cuda:0"
    ref = _pooled_model(dev)
    ref.compile("adagrad", "binary_crossentropy", metrics=[])
    ref.train()
    ref_loss = []
    for step in range(STEPS):
        Xg, yg = _pooled_batch(step, world)
        ref_loss.append(ref._train_step(Xg.to(dev), yg.to(dev))[0].reshape(()))
    torch.cuda.synchronize()
    got_loss = sum(r["loss"] for r in ranks)
    assert torch.allclose(got_loss, torch.stack(ref_loss).cpu(), rtol=2e-5), (got_loss, ref_loss)
    for k, v in ref.state_dict().items():
        v = v.detach().cpu()
        err = float((ranks[0]["sd"][k] - v).abs().max())
        assert err <= 2e-6 * max(1.0, float(v.abs().max())) + 2e-7, "%s: %.3e" % (k, err)
