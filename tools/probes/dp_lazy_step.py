#!/usr/bin/env python
"""Step time of the replicated-tables trainer on the reference's default kwargs (L2 on the tables, adam) at one rank: the lazy
update it keeps since round 6 against the dense O(vocabulary) route (DCTR_DP_LAZY=0).  Criteo shape, batch 4096.
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/probes/dp_lazy_step.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.argv = ["bench.py"]
import torch, torch.distributed as dist
import bench as b
from deepctr_torch import parallel as par
args = b.parse()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
X, y = b.synth(args, "cuda:0", 0)
B = args.batch
for mode in ("1", "0"):
    os.environ["DCTR_DP_LAZY"] = mode
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("C%d" % (i + 1), args.vocab, b.DIM) for i in range(b.F_SPARSE)] + \
           [DenseFeat("I%d" % (i + 1), 1) for i in range(b.N_DENSE)]
    model = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=1e-5, l2_reg_embedding=1e-5, dnn_dropout=0,
                   seed=1024, device="cuda:0")
    model.compile("adam", "binary_crossentropy", metrics=[])
    model.train()
    tr = par.DataParallelTrainer(model)
    n = 64 if mode == "1" else 6
    for i in range(3):
        tr.train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        j = (3 + i) % (X.shape[0] // B)
        tr.train_step(X[j * B:(j + 1) * B], y[j * B:(j + 1) * B])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("DCTR_DP_LAZY=%s  lazy=%s  %.3f ms per step (eager, host-paced)" % (mode, tr._lazy, dt * 1e3))
    tr.close(); del tr, model; torch.cuda.empty_cache()
dist.destroy_process_group()
