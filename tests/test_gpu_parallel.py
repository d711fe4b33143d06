"""GPU: DataParallelTrainer with a 1-rank RCCL group drives the real kernels (gather -> exchange -> global
deterministic update) and must reproduce the single-GPU fused step bit for bit; the multi-rank algebra is
covered on CPU by tests/test_parallel_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import build_model, load_golden, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_trainer_matches_single_gpu_step_and_reference(one_rank_group, opt):
    from deepctr_torch.parallel import DataParallelTrainer
    g = load_golden("deepfm_criteo")
    models = []
    for use_trainer in (False, True):
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        tr = DataParallelTrainer(m) if use_trainer else None
        losses = []
        for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
            xb, yb = torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV)
            loss = (tr.train_step(xb, yb) if tr else m._train_step(xb, yb))[0]
            losses.append(loss.item())
        if tr:
            tr.close()
        np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=2e-5)
        models.append(m)
    a, b = models[0].state_dict(), models[1].state_dict()
    for k in a:
        # same kernels, same (id, sample) summation order; only FM's fold is done by torch ops in the trainer
        assert max_abs(a[k].cpu().numpy(), b[k].cpu().numpy()) <= 1e-6, k
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            assert max_abs(b[k[len(opt) + 2:]].cpu().numpy(), v) <= 2e-5, k


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
@pytest.mark.parametrize("graphs", [False, True])
def test_sharded_trainer_one_rank_matches_single_gpu_and_reference(one_rank_group, opt, graphs):
    """ShardedTrainer with a 1-rank RCCL group drives the real kernels of the table-sharded exchange (pack ids ->
    all-to-all -> owner gather -> all-to-all -> assemble -> tower / head -> assemble^T -> all-to-all -> owner update,
    dense all-reduce, slab step), eagerly and as five hipGraph segments, and must reproduce the single-GPU fused
    step and the reference trajectory.  The multi-rank algebra is covered on CPU by tests/test_sharded_gloo.py."""
    from deepctr_torch.parallel import ShardedTrainer
    g = load_golden("deepfm_criteo")
    models = []
    for use_trainer in (False, True):
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        tr = ShardedTrainer(m, use_graphs=graphs) if use_trainer else None
        losses = []
        steps = list(zip(g["extra"]["X_steps"], g["extra"]["y_steps"]))
        dev_steps = [(torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV)) for Xb, yb in steps]
        for k, (xb, yb) in enumerate(dev_steps):
            if tr:   # announce the next batch on the first step only: both id routes are exercised
                nxt = dev_steps[k + 1][0] if k == 0 else None
                loss = tr.train_step(xb, yb, next_xb=nxt)[0]
            else:
                loss = m._train_step(xb, yb)[0]
            losses.append(loss.item())
        if tr:
            tr.gather_tables()
            tr.close()
            m.model_plan().check_ids()
        np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=2e-5)
        models.append(m)
    a, b = models[0].state_dict(), models[1].state_dict()
    for k in a:
        assert max_abs(a[k].cpu().numpy(), b[k].cpu().numpy()) <= 2e-6, k
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            assert max_abs(b[k[len(opt) + 2:]].cpu().numpy(), v) <= 2e-5, k
    # the model predicts normally again once the trainer is closed
    models[1].eval()
    with torch.no_grad():
        y = models[1](torch.from_numpy(g["X"]).to(DEV))
    assert y.shape[0] == g["X"].shape[0]


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_trainer_runs_pooled_and_shared_tables(one_rank_group, opt):
    """DataParallelTrainer on general update units (round 5): tests/golden/deepfm_mixed -- sum / mean / max pooled VarLen
    fields, a history sharing the item table, a length column -- through gather -> all-gather of row gradients and arg-max
    positions -> the global sorted update, against the reference's 3-step trajectory and the single-GPU step."""
    from deepctr_torch.parallel import DataParallelTrainer
    g = load_golden("deepfm_mixed")
    models = []
    for use_trainer in (False, True):
        m = build_model(g["spec"], DEV)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        assert m.model_plan().gen is not None
        tr = DataParallelTrainer(m) if use_trainer else None
        losses = []
        for Xb, yb in zip(g["extra"]["X_steps"], g["extra"]["y_steps"]):
            xb, yb = torch.from_numpy(Xb).to(DEV), torch.from_numpy(yb).to(DEV)
            losses.append((tr.train_step(xb, yb) if tr else m._train_step(xb, yb))[0].item())
        if tr:
            tr.close()
        torch.cuda.synchronize()
        m.model_plan().check_ids()
        np.testing.assert_allclose(losses, g["extra"][opt + "3_loss"], rtol=2e-5)
        models.append(m)
    a, b = models[0].state_dict(), models[1].state_dict()
    for k in a:
        assert max_abs(a[k].cpu().numpy(), b[k].cpu().numpy()) <= 2e-6, k
    for k, v in g["extra"].items():
        if k.startswith(opt + "3/"):
            assert max_abs(b[k[len(opt) + 2:]].cpu().numpy(), v) <= 2e-5, k
