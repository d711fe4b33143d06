#!/usr/bin/env python
"""A few eager train steps of one model at the Criteo shape (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch.inputs import DenseFeat, SparseFeat  # noqa: E402
from deepctr_torch import models as M  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "DCN"
dev, B, V = "cuda:0", 4096, 1_000_000
cols = [SparseFeat("C%d" % i, V, 16) for i in range(26)] + [DenseFeat("I%d" % i, 1) for i in range(13)]
kw = dict(l2_reg_linear=0, l2_reg_embedding=0, device=dev)
if name == "DeepFM":
    m = M.DeepFM(cols, cols, dnn_hidden_units=(256, 128), **kw)
elif name == "DCN":
    m = M.DCN(cols, cols, dnn_hidden_units=(256, 128), **kw)
elif name == "xDeepFM":
    m = M.xDeepFM(cols, cols, dnn_hidden_units=(256, 256), cin_layer_size=(128, 128), **kw)
elif name == "FiBiNET":
    m = M.FiBiNET(cols, cols, dnn_hidden_units=(128, 128), **kw)
else:
    m = M.PNN(cols, dnn_hidden_units=(256, 128), l2_reg_embedding=0, device=dev)
m.compile("adagrad", "binary_crossentropy", metrics=[])
m.train()
g = torch.Generator().manual_seed(0)
X = torch.cat([torch.randint(0, V, (B, 26), generator=g).float(), torch.rand(B, 13, generator=g)], 1).to(dev)
y = torch.randint(0, 2, (B,), generator=g).float().to(dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    m._train_step(X, y)
torch.cuda.synchronize()
