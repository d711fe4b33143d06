#!/usr/bin/env python
"""Which Python lines of one eager train step launch ATen fill / copy / add / mm kernels (Python wrappers around the tensor methods that log their callers).
    python tools/probes/aten_in_step.py xDeepFM"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402


from deepctr_torch import models as M  # noqa: E402
from deepctr_torch.inputs import DenseFeat, SparseFeat  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "xDeepFM"
dev, B, V = "cuda:0", 4096, 1_000_000
cols = [SparseFeat("C%d" % i, V, 16) for i in range(26)] + [DenseFeat("I%d" % i, 1) for i in range(13)]
kw = dict(l2_reg_linear=0, l2_reg_embedding=0, device=dev)
if name == "xDeepFM":
    m = M.xDeepFM(cols, cols, dnn_hidden_units=(256, 256), cin_layer_size=(128, 128), **kw)
elif name == "FiBiNET":
    m = M.FiBiNET(cols, cols, dnn_hidden_units=(128, 128), **kw)
else:
    m = getattr(M, name)(cols, cols, dnn_hidden_units=(256, 128), **kw)
m.compile("adagrad", "binary_crossentropy", metrics=[])
m.train()
g = torch.Generator().manual_seed(0)
X = torch.cat([torch.randint(0, V, (B, 26), generator=g).float(), torch.rand(B, 13, generator=g)], 1).to(dev)
y = torch.randint(0, 2, (B,), generator=g).float().to(dev)
for _ in range(4):
    m._train_step(X, y)
torch.cuda.synchronize()
import traceback  # noqa: E402

LOG = []


def _wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if "deepctr_torch" in x.filename][-3:]
        shp = tuple(a[0].shape) if (a and torch.is_tensor(a[0])) else (a[0] if a else None)
        LOG.append("%-12s %-22s %s" % (name, str(shp)[:22], " <- ".join("%s:%d %s" % (os.path.basename(x.filename), x.lineno, x.name)
                                                                       for x in reversed(fr))))
        return orig(*a, **k)
    setattr(owner, name, f)


for nm in ("zero_", "copy_", "add_", "fill_", "contiguous", "clone"):
    _wrap(torch.Tensor, nm)
for nm in ("zeros", "zeros_like", "cat", "mm"):
    _wrap(torch, nm)
m._train_step(X, y)
torch.cuda.synchronize()
print("\n".join(LOG))
