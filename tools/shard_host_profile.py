#!/usr/bin/env python
"""Host-side cost of every statement of ShardedTrainer.train_step at 1 rank (the multi-GPU step is host-bound)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
args = bench.parse()
model = bench.build_model(args, "cuda:0")
X, y = bench.synth(args, "cuda:0", 0)
from deepctr_torch import parallel as par  # noqa: E402

tr = par.ShardedTrainer(model, use_graphs=True)
B = args.batch
nb = X.shape[0] // B


def batch(i):
    j = i % nb
    return X[j * B:(j + 1) * B], y[j * B:(j + 1) * B]


for i in range(8):
    tr.train_step(*batch(i), next_xb=batch(i + 1)[0])
torch.cuda.synchronize()

# re-implement train_step with timers (same statements)
acc = {}


def T(name, fn):
    t0 = time.perf_counter()
    r = fn()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
    return r


n = 300
lay = tr.layout
t_all = time.perf_counter()
for i in range(8, 8 + n):
    xb, yb = batch(i)
    nxt = batch(i + 1)[0]
    T("copy_x", lambda: tr._x.copy_(xb))
    T("copy_y", lambda: tr._y.copy_(yb))
    T("pack_next_ids", lambda: tr._ids_next.copy_(tr.ops.pack_ids(nxt)))
    chunks, tr._ids_t = T("segB", tr._segB)
    T("a2a_rows", lambda: dist.all_to_all_single(tr._recv, chunks))
    send, loss, y_pred = T("segC", tr._segC)
    T("a2a_grads", lambda: dist.all_to_all_single(tr._grads_all, send))
    work = T("allreduce_async", lambda: dist.all_reduce(tr.slab.grad, async_op=True))
    T("segD", tr._segD)
    T("wait", work.wait)
    T("segE", tr._segE)
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / n * 1e6
print("us per step (host-paced): %.1f" % total)
for k, v in acc.items():
    print("  %-16s %6.1f us" % (k, v / n * 1e6))
dist.destroy_process_group()
