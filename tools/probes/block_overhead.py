#!/usr/bin/env python
"""Where the time of ONE block of the sharded step goes (one rank, direct exchange): host time of the pieces of
ShardedTrainer.train_block against the block's wall time between two synchronisations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
import torch
import torch.distributed as dist
sys.argv = ["bench.py"]
import bench as b
args = b.parse()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from deepctr_torch import parallel as par
model = b.build_model(args, "cuda:0")
X, y = b.synth(args, "cuda:0", 0)
B, S = args.batch, int(os.environ.get("S", "20"))
tr = par.ShardedTrainer(model, use_graphs=False, exchange="direct")
for k in range(3):
    tr.train_step(X[k * B:(k + 1) * B], y[k * B:(k + 1) * B], next_xb=X[(k + 1) * B:(k + 2) * B])
tr.set_use_graphs(True)
def blk(j):
    return X[j * B:(j + S) * B].view(S, B, X.shape[1]), y[j * B:(j + S) * B].view(S, B)
pos = 0
def run():
    global pos
    j = pos
    jn = j + S if j + 2 * S <= X.shape[0] // B else 0
    xs, ys = blk(j)
    pos = jn
    return tr.train_block(xs, ys, next_first=X[jn * B:(jn + 1) * B])
for _ in range(4):
    run()
torch.cuda.synchronize()
# instrument graph replay
import deepctr_torch.parallel as P
orig = torch.cuda.CUDAGraph.replay
acc = {"replay": 0.0}
def timed(self):
    t = time.perf_counter(); orig(self); acc["replay"] += time.perf_counter() - t
torch.cuda.CUDAGraph.replay = timed
for n_blocks in (1, 4):
    for rep in range(3):
        acc["replay"] = 0.0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n_blocks):
            run()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("blocks %d of %d steps: host %.0f us (graph.replay %.0f us), wall %.0f us = %.1f us/step" % (
            n_blocks, S, (t1 - t0) * 1e6, acc["replay"] * 1e6, (t2 - t0) * 1e6, (t2 - t0) * 1e6 / (n_blocks * S)), flush=True)
dist.destroy_process_group()
