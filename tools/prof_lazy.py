#!/usr/bin/env python
"""A few train steps of DeepFM with the reference's default kwargs (lazy exact update) for rocprofv3 --stats."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch.inputs import DenseFeat, SparseFeat  # noqa: E402
from deepctr_torch import models as M  # noqa: E402

dev, B, V = "cuda:0", 4096, 1_000_000
cols = [SparseFeat("C%d" % i, V, 16) for i in range(26)] + [DenseFeat("I%d" % i, 1) for i in range(13)]
m = M.DeepFM(cols, cols, dnn_hidden_units=(256, 128), device=dev)
m.compile(sys.argv[1] if len(sys.argv) > 1 else "adam", "binary_crossentropy", metrics=[])
m.train()
g = torch.Generator().manual_seed(0)
n = B * 8
X = torch.cat([torch.randint(0, V, (n, 26), generator=g).float(), torch.rand(n, 13, generator=g)], 1).to(dev)
y = torch.randint(0, 2, (n,), generator=g).float().to(dev)
for i in range(4):
    m._train_step(X[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    m._train_step(X[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B])
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3, file=sys.stderr)
