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
