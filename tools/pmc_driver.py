#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (tools/pmc_traffic.sh): a few launches of the hand-written embedding
kernels at the bench shape plus three calibration kernels whose HBM byte counts are known, so that FETCH_SIZE /
WRITE_SIZE can be corrected as MI355X_MICROARCH.md (HBM section) prescribes before they are compared with
algorithmic bytes.

calibration launches (ATen kernels, 256 MiB operands -- larger than the 256 MiB Infinity Cache when summed):
  stream   torch.add(a, 1, out=b)            reads 256 MiB, writes 256 MiB, 16 B / lane coalesced
  reduce   a.sum()                           reads 256 MiB
  gather   torch.index_select(T[4M,16], idx) reads 1 Mi random 64-byte rows (64 MiB) + 8 MiB of indices, writes 64 MiB
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
n = 64 * 1024 * 1024
a = torch.rand(n, device=dev)
b = torch.empty_like(a)
T = torch.rand(4 * 1024 * 1024, 16, device=dev)
idx = torch.randint(0, T.shape[0], (1024 * 1024,), device=dev)
for _ in range(3):
    torch.add(a, 1.0, out=b)
    a.sum()
    torch.index_select(T, 0, idx)
torch.cuda.synchronize()
del a, b, T, idx
torch.cuda.empty_cache()


class A:
    vocab, batch, optimizer = 1_000_000, 4096, "adagrad"


# (PMC_OPTS / PMC_BATCHES restrict the sweep: the bench line needs adagrad at 4096 only)
for opt in os.environ.get("PMC_OPTS", "adagrad,sgd").split(","):
    A.optimizer = opt
    model = bench.build_model(A, dev)
    for Bsz in [int(b) for b in os.environ.get("PMC_BATCHES", "4096,32768").split(",")]:
        gen = torch.Generator().manual_seed(0)
        X = torch.cat([torch.randint(0, A.vocab, (8 * Bsz, 26), generator=gen).float(),
                       torch.rand(8 * Bsz, 13, generator=gen)], 1).to(dev)
        bench.time_hot_kernels(model, X, Bsz, 5, opt, ring=8)
    del model
    torch.cuda.empty_cache()
torch.cuda.synchronize()
