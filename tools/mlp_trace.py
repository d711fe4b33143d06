#!/usr/bin/env python
"""Per-workgroup phase timeline of the tower kernels at the DeepFM shape (B=4096, 429 -> 256 -> 128 -> 1)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch._hip import lib as L  # noqa: E402

L.use_diag_library()   # make -C deepctr-torch_amd/csrc diag
from deepctr_torch._hip import mlp  # noqa: E402
from deepctr_torch.layers import DNN  # noqa: E402

dev = "cuda:0"
B = 4096
dnn = DNN(429, (256, 128), init_std=0.05, device=dev)
lin = torch.nn.Linear(128, 1, bias=False).to(dev)
xs = [torch.randn(B, 432, device=dev) for _ in range(4)]
lib = L.lib()


def step(i):
    x = xs[i % 4].clone().requires_grad_(True)
    y = mlp.tower(dnn, lin, x, 429)
    y.backward(torch.ones_like(y))


for i in range(3):
    step(i)
torch.cuda.synchronize()
buf = torch.zeros(3 * 4096 * 16, dtype=torch.int64, device=dev)
lib.dctr_dbg_mlp_trace(ctypes.c_void_p(buf.data_ptr()))
step(3)
torch.cuda.synchronize()
lib.dctr_dbg_mlp_trace(None)
t = buf.view(3, 4096, 16).cpu().numpy().astype(np.int64)
tick = 0.01  # us


def st(v):
    v = np.asarray(v, dtype=np.float64) * tick
    return {"mean": round(float(v.mean()), 2), "p10": round(float(np.percentile(v, 10)), 2),
            "p50": round(float(np.percentile(v, 50)), 2), "p90": round(float(np.percentile(v, 90)), 2),
            "max": round(float(v.max()), 2)}


res = {}
f = t[0][t[0][:, 0] > 0]
t0 = f[:, 0].min()
res["fwd"] = {"n_wg": int(len(f)), "start": st(f[:, 0] - t0), "stage_x": st(f[:, 1] - f[:, 0]),
              "layer0_mfma": st(f[:, 2] - f[:, 1]), "layer0_epilogue": st(f[:, 3] - f[:, 2]),
              "layer0_barrier": st(f[:, 4] - f[:, 3]), "layer1_mfma": st(f[:, 5] - f[:, 4]),
              "layer1_epilogue": st(f[:, 6] - f[:, 5]), "layer1_barrier": st(f[:, 7] - f[:, 6]),
              "projection": st(f[:, 15] - f[:, 7]), "wg_total": st(f[:, 15] - f[:, 0]), "end": st(f[:, 15] - t0)}
b = t[1][t[1][:, 0] > 0]
t0 = b[:, 0].min()
res["bwd_data"] = {"n_wg": int(len(b)), "start": st(b[:, 0] - t0), "stage_top": st(b[:, 1] - b[:, 0]),
                   "layer1_mfma+epi": st(b[:, 2] - b[:, 1]), "layer1_barrier": st(b[:, 3] - b[:, 2]),
                   "layer0_mfma+epi": st(b[:, 4] - b[:, 3]), "layer0_barrier": st(b[:, 5] - b[:, 4]),
                   "wg_total": st(b[:, 15] - b[:, 0]), "end": st(b[:, 15] - t0)}
w = t[2][t[2][:, 0] > 0]
t0 = w[:, 0].min()
wk = w[w[:, 1] > 0]
res["wgrad"] = {"n_wg": int(len(w)), "n_gemm_wg": int(len(wk)), "start": st(wk[:, 0] - t0),
                "mainloop": st(wk[:, 1] - wk[:, 0]), "lds_park+barrier": st(wk[:, 2] - wk[:, 1]),
                "combine+store": st(wk[:, 3] - wk[:, 2]), "wg_total": st(wk[:, 3] - wk[:, 0]), "end": st(wk[:, 3] - t0),
                "by_layer_total": {int(l): st(wk[wk[:, 14] == l][:, 3] - wk[wk[:, 14] == l][:, 0]) for l in set(wk[:, 14])}}
print(json.dumps(res, indent=1))
