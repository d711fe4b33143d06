#!/usr/bin/env python
"""Event-time the hand-written embedding kernels at the bench shape (dev loop helper)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd")); sys.path.insert(0, ROOT)
import torch
import bench


class A:
    vocab, batch, optimizer, ids = 1_000_000, 4096, "adagrad", "uniform"


for opt in sys.argv[1:] or ["adagrad"]:
    A.optimizer = opt
    model = bench.build_model(A, "cuda:0")
    for Bsz in (4096, 32768):
        gen = torch.Generator().manual_seed(0)
        X = torch.cat([torch.randint(0, A.vocab, (Bsz, 26), generator=gen).float(), torch.rand(Bsz, 13, generator=gen)], 1).to("cuda:0")
        k = bench.time_hot_kernels(model, X, Bsz, 30, opt)
        alg = bench.algorithmic_bytes(Bsz, opt)
        print(opt, Bsz, {n: "%.1fus %.0fGB/s" % (v["min_us"], alg[n] / v["min_us"] / 1e3) for n, v in k.items()})
