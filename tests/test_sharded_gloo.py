"""CPU, gloo: table-sharded training (deepctr_torch.parallel.ShardedTrainer) is correct by construction.

The REAL trainer, the REAL drop-in DeepFM, DenseSlab and the tower / head autograd Functions run in every process;
only the device kernels are stood in for -- the tower / head / dense optimizer by tests/mock_lib.py, the four
embedding steps of the exchange by tests/shard_standin.py (same layouts, torch index ops).  After a few steps and
``gather_tables()`` every rank must hold the parameters ONE process reaches on the concatenated batch with the
reference's dense-gradient algorithm (oracle/torch_port.py, pinned to the reference's golden vectors).
World sizes 2 and 3 over 5 fields: owners with 3/2 and 2/2/1 units (an unused chunk slot on some ranks)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F_, V_, D_, ND_, B_ = 5, 30, 8, 3, 16


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


def _model():
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("C%d" % i, V_, D_) for i in range(F_)] + [DenseFeat("I%d" % i, 1) for i in range(ND_)]
    return DeepFM(cols, cols, dnn_hidden_units=(16, 8), l2_reg_linear=0, l2_reg_embedding=0, init_std=0.1, seed=7,
                  device="cpu")


def _patch_for_cpu():
    """The product path refuses CPU tensors; the stand-ins take its place in this process."""
    from deepctr_torch._hip import lib as L
    from mock_lib import MockLib
    m = MockLib()
    L.lib = lambda: m
    L.require_gpu = lambda t, what: None
    L.stream_handle = lambda device=None: None
    torch.Tensor.is_cuda = property(lambda self: True)


def _worker(rank, world, port, opt_name, out_dir):
    for p in (os.path.join(ROOT, "deepctr-torch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        _patch_for_cpu()
        from deepctr_torch.parallel import ShardLayout, ShardedTrainer
        from shard_standin import TorchShardOps
        m = _model()
        if rank != 0:                     # broadcast_parameters must make the replicas identical
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01)
        m.compile(opt_name.split("+")[0], "binary_crossentropy", metrics=[])
        m.train()
        lay = ShardLayout(m.model_plan(), world, rank)
        tr = ShardedTrainer(m, ops=TorchShardOps(m, lay))
        assert tr.layout.n_slots == (F_ + world - 1) // world
        batches = [_batch(step, world) for step in range(4)]
        mine = [(Xg[rank * B_:(rank + 1) * B_].contiguous(), yg[rank * B_:(rank + 1) * B_].contiguous()) for Xg, yg in batches]
        use_block = opt_name.endswith("+block")
        for step in range(2 if use_block else 4):
            # steps 0 -> 1 and 2 -> 3 announce the next batch (ids ride in the gradient all-to-all); step 2 does not
            nxt = mine[step + 1][0] if step in (0, 2) else None
            tr.train_step(mine[step][0], mine[step][1], next_xb=nxt)
        if use_block:     # steps 2 and 3 as ONE train_block call (without the direct exchange: the same two train_steps)
            tr.train_block(torch.stack([mine[2][0], mine[3][0]]), torch.stack([mine[2][1], mine[3][1]]), next_first=None)
        tr.gather_tables()
        tr.close()
        torch.save({k: v.detach().clone() for k, v in m.state_dict().items()}, os.path.join(out_dir, "rank%d.pt" % rank))
        if rank == 0:
            torch.save({k: v.detach().clone() for k, v in _model().state_dict().items()}, os.path.join(out_dir, "init.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("opt_name", ["sgd", "adagrad", "adagrad+block"])
def test_sharded_training_equals_single_process_on_the_global_batch(tmp_path, opt_name, world):
    if opt_name.endswith("+block") and world == 3:
        pytest.skip("one train_block case is enough")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, opt_name, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for r in range(1, world):
        for k in ranks[0]:
            assert torch.equal(ranks[0][k], ranks[r][k]), "replicas differ after gather_tables: %s" % k
    for p in (os.path.join(ROOT, "oracle"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    from torch_port import DeepFMPort, train_step
    torch.set_num_threads(1)
    init = torch.load(os.path.join(str(tmp_path), "init.pt"))
    ref = DeepFMPort(F_, V_, D_, ND_, hidden=(16, 8))
    ref.load_reference_state({k: v.numpy() for k, v in init.items()}, ["C%d" % i for i in range(F_)])
    opt = (torch.optim.SGD(ref.parameters(), lr=0.01) if opt_name == "sgd" else torch.optim.Adagrad(ref.parameters()))   # ("adagrad+block": Adagrad)
    for step in range(4):
        Xg, yg = _batch(step, world)
        train_step(ref, opt, Xg, yg)
    got = ranks[0]
    pairs = [("out.bias", ref.bias), ("linear_model.weight", ref.lin_w), ("dnn_linear.weight", ref.dnn_linear.weight)]
    for i, l in enumerate(ref.linears):
        pairs += [("dnn.linears.%d.weight" % i, l.weight), ("dnn.linears.%d.bias" % i, l.bias)]
    for f in range(F_):
        pairs += [("embedding_dict.C%d.weight" % f, ref.emb[f].weight),
                  ("linear_model.embedding_dict.C%d.weight" % f, ref.lin[f].weight)]
    assert len(pairs) == len(got)
    for k, v in pairs:
        err = float((got[k] - v.detach()).abs().max())
        assert err <= 2e-5 * max(1.0, float(v.abs().max())), "%s: %.3e" % (k, err)
    moved = float((got["embedding_dict.C0.weight"] - init["embedding_dict.C0.weight"]).abs().max())
    assert moved > 1e-4, "the tables did not train"


def test_shard_layout():
    sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
    from deepctr_torch.parallel import ShardLayout
    m = _model()
    plan = m.model_plan()
    lay = ShardLayout(plan, 8, 3)
    assert lay.F == 5 and lay.n_slots == 1 and lay.owned == [3] and lay.wide_col == 8 and lay.ids_col == 12 and lay.ldc == 16
    lay = ShardLayout(plan, 2, 1)
    assert lay.owned == [1, 3] and lay.n_slots == 3 and lay.wide_col == 24 and lay.ids_col == 28 and lay.ldc == 32
    assert len(lay.id_cols) == 2 * 3 and lay.id_cols[:3] == [0, 2, 4] and lay.id_cols[3:5] == [1, 3]


# ---- models outside the fused train step (round 3): xDeepFM, FiBiNET, DCN shard their tables too -----------------------
def _other_model(kind):
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch import models as M
    # unequal vocabularies: the byte-balanced owner map differs from round-robin
    cols = [SparseFeat("C%d" % i, V_ + 7 * ((3 * i) % F_), D_) for i in range(F_)] + [DenseFeat("I%d" % i, 1) for i in range(ND_)]
    kw = dict(l2_reg_linear=0, l2_reg_embedding=0, init_std=0.1, seed=11, device="cpu")
    if kind == "xdeepfm":
        return M.xDeepFM(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(6, 4), l2_reg_dnn=1e-2, l2_reg_cin=1e-2, **kw)
    if kind == "fibinet":
        return M.FiBiNET(cols, cols, dnn_hidden_units=(8,), bilinear_type="interaction", reduction_ratio=2, **kw)
    # (DCN keeps the reference's quirk of regularising the linear tables whatever l2_reg_linear says -> the lazy update,
    # which is single-GPU; PNN has no linear model at all: the "none" case of the wide half)
    kw.pop("l2_reg_linear")
    return M.PNN(cols, dnn_hidden_units=(8,), use_inner=True, use_outter=False, l2_reg_dnn=1e-2, **kw)


def _other_batch(step, world):
    g = torch.Generator().manual_seed(500 + step)
    ids = torch.randint(0, V_, (world * B_, F_), generator=g).float()
    X = torch.cat([ids, torch.rand(world * B_, ND_, generator=g)], 1)
    y = torch.randint(0, 2, (world * B_,), generator=g).float()
    return X, y


def _other_worker(rank, world, port, kind, out_dir):
    for p in (os.path.join(ROOT, "deepctr-torch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        _patch_for_cpu()
        from deepctr_torch.parallel import ShardLayout, ShardedTrainer
        from shard_standin import TorchShardOps
        m = _other_model(kind)
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        lay = ShardLayout(m.model_plan(), world, rank)
        tr = ShardedTrainer(m, ops=TorchShardOps(m, lay))
        # (xDeepFM / FiBiNET are outside the fused step; PNN is inside but its L2 on the tower sends it here too)
        assert tr.slab is None and tr.bucket is not None, "%s is expected on the autograd route" % kind
        for step in range(3):
            Xg, yg = _other_batch(step, world)
            nxt = _other_batch(step + 1, world)[0][rank * B_:(rank + 1) * B_].contiguous() if step == 0 else None
            tr.train_step(Xg[rank * B_:(rank + 1) * B_].contiguous(), yg[rank * B_:(rank + 1) * B_].contiguous(), next_xb=nxt)
        tr.gather_tables()
        tr.close()
        torch.save({k: v.detach().clone() for k, v in m.state_dict().items()}, os.path.join(out_dir, "rank%d.pt" % rank))
        if rank == 0:
            torch.save(lay.owner, os.path.join(out_dir, "owner.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("xdeepfm", 2), ("fibinet", 2), ("pnn", 3)])
def test_sharded_training_of_autograd_route_models(tmp_path, kind, world, monkeypatch):
    mp.spawn(_other_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for r in range(1, world):
        for k in ranks[0]:
            assert torch.equal(ranks[0][k], ranks[r][k]), "replicas differ after gather_tables: %s" % k
    owner = torch.load(os.path.join(str(tmp_path), "owner.pt"))
    counts = [owner.count(q) for q in range(world)]
    assert max(counts) - min(counts) <= 1
    assert owner != [u % world for u in range(F_)], "the byte-balanced map should differ from round-robin here"
    # one process, the global batch, the same drop-in model
    for p in (os.path.join(ROOT, "deepctr-torch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from deepctr_torch._hip import lib as L
    from mock_lib import MockLib
    mk = MockLib()
    monkeypatch.setattr(L, "lib", lambda: mk)
    monkeypatch.setattr(L, "require_gpu", lambda t, what: None)
    monkeypatch.setattr(L, "stream_handle", lambda device=None: None)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        m = _other_model(kind)
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()
        for step in range(3):
            Xg, yg = _other_batch(step, world)
            m._train_step(Xg, yg)
    finally:
        torch.set_num_threads(threads)
    for k, v in m.state_dict().items():
        err = float((ranks[0][k] - v.detach()).abs().max())
        assert err <= 2e-5 * max(1.0, float(v.abs().max())), "%s: %.3e" % (k, err)
