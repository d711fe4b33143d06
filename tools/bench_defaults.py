#!/usr/bin/env python
"""Train-step time with the reference's DEFAULT kwargs (l2_reg_embedding = l2_reg_linear = 1e-5, compile('adam')) at
the Criteo shape: the exact lazy update (csrc/lazy.hip, O(batch)) against the exact dense path (DCTR_LAZY_UPDATE=0:
dense [V, D] gradients + torch.optim over 442 M table parameters, what the reference does).
    python tools/bench_defaults.py > gpurun_out/defaults.json"""
import json
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
gen = torch.Generator().manual_seed(0)
n = B * 16
X = torch.cat([torch.randint(0, V, (n, 26), generator=gen).float(), torch.rand(n, 13, generator=gen)], 1).to(dev)
y = torch.randint(0, 2, (n,), generator=gen).float().to(dev)
CASES = [
    ("DeepFM defaults (l2=1e-5, adam)", lambda: M.DeepFM(cols, cols, dnn_hidden_units=(256, 128), device=dev), "adam"),
    ("DeepFM l2=1e-5, adagrad", lambda: M.DeepFM(cols, cols, dnn_hidden_units=(256, 128), device=dev), "adagrad"),
    ("DCN defaults (adagrad)", lambda: M.DCN(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0,
                                              device=dev), "adagrad"),
]
res = {}
for name, make, opt in CASES:
    for lazy in ("1", "0"):
        os.environ["DCTR_LAZY_UPDATE"] = lazy
        try:
            m = make()
            m.compile(opt, "binary_crossentropy", metrics=[])
            m.train()
            steps = 24 if lazy == "1" else 6

            def batch(i):
                j = i % 16
                return X[j * B:(j + 1) * B], y[j * B:(j + 1) * B]

            for i in range(3):
                m._train_step(*batch(i))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                m._train_step(*batch(3 + i))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            r = {"mode": m.model_plan().update[0], "fused_step": bool(m._fused and m._fused.get("ok")),
                 "ms_per_step": dt * 1e3, "samples_per_s": B / dt}
            if lazy == "1" and r["fused_step"]:
                try:                                # hipGraph replay of the fused lazy step (what fit() does)
                    from deepctr_torch._hip.graph import GraphedTrainStep
                    gs = GraphedTrainStep(m, *batch(0), steps_per_graph=4, inputs_ready=True).capture(*batch(0))
                    for i in range(8):
                        gs(*batch(i))
                    gs.flush()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(96):
                        gs(*batch(i))
                    gs.flush()
                    torch.cuda.synchronize()
                    gt = (time.perf_counter() - t0) / 96
                    r.update(graph_ms_per_step=gt * 1e3, graph_samples_per_s=B / gt)
                except Exception as exc:  # noqa: BLE001
                    r["graph_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
                    torch.cuda.synchronize()
            t1 = time.perf_counter()
            m.state_dict()                         # includes the flush of every row in lazy mode
            torch.cuda.synchronize()
            r["state_dict_ms"] = (time.perf_counter() - t1) * 1e3
            m.model_plan().check_ids()
            res["%s | lazy=%s" % (name, lazy)] = r
            del m
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            res["%s | lazy=%s" % (name, lazy)] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
print(json.dumps(res, indent=1))
