#!/usr/bin/env python
"""Train-step time of the other BASELINE.json configurations (parity-test cases, not bench lines): xDeepFM
(CIN [128,128]), FiBiNET (bilinear 'interaction' + SENET), DCN, PNN at the Criteo shape, batch 4096, Adagrad, l2=0.
Eager steps and one-hipGraph-per-step replays.    python tools/bench_models.py > gpurun_out/models.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch.inputs import DenseFeat, SparseFeat  # noqa: E402
from deepctr_torch import models as M  # noqa: E402
from deepctr_torch._hip.graph import GraphedTrainStep  # noqa: E402

dev, B, V = "cuda:0", 4096, 1_000_000
cols = [SparseFeat("C%d" % i, V, 16) for i in range(26)] + [DenseFeat("I%d" % i, 1) for i in range(13)]
gen = torch.Generator().manual_seed(0)
n = B * 16
X = torch.cat([torch.randint(0, V, (n, 26), generator=gen).float(), torch.rand(n, 13, generator=gen)], 1).to(dev)
y = torch.randint(0, 2, (n,), generator=gen).float().to(dev)
spec = {
    "DeepFM": lambda: M.DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0, device=dev),
    "xDeepFM": lambda: M.xDeepFM(cols, cols, dnn_hidden_units=(256, 256), cin_layer_size=(128, 128), cin_split_half=True,
                                 l2_reg_linear=0, l2_reg_embedding=0, device=dev),
    "FiBiNET": lambda: M.FiBiNET(cols, cols, dnn_hidden_units=(128, 128), l2_reg_linear=0, l2_reg_embedding=0, device=dev),
    "DCN": lambda: M.DCN(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0, device=dev),
    "DCN_matrix": lambda: M.DCN(cols, cols, cross_parameterization="matrix", dnn_hidden_units=(256, 128), l2_reg_linear=0,
                                l2_reg_embedding=0, l2_reg_cross=0, device=dev),
    "PNN": lambda: M.PNN(cols, dnn_hidden_units=(256, 128), l2_reg_embedding=0, device=dev),
    "NFM": lambda: M.NFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0, device=dev),
    "WDL": lambda: M.WDL(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0, device=dev),
    "AutoInt": lambda: M.AutoInt(cols, cols, att_layer_num=3, att_head_num=2, dnn_hidden_units=(256, 128),
                                 l2_reg_embedding=0, device=dev),
    "DCNMix": lambda: M.DCNMix(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0,
                               l2_reg_cross=0, device=dev),
    "AFM": lambda: M.AFM(cols, cols[:26], attention_factor=8, l2_reg_linear=0, l2_reg_embedding=0, l2_reg_att=0,
                         device=dev),
}
res = {}
only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else None
for name, make in spec.items():
    if only is not None and name not in only:
        continue
    try:
        m = make()
        m.compile("adagrad", "binary_crossentropy", metrics=[])
        m.train()

        def batch(i):
            j = i % 16
            return X[j * B:(j + 1) * B], y[j * B:(j + 1) * B]

        for i in range(3):
            m._train_step(*batch(i))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            m._train_step(*batch(i))
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20
        r = {"eager_ms": eager * 1e3, "eager_samples_per_s": B / eager, "fused_step": bool(m._fused and m._fused.get("ok"))}
        try:
            gs = GraphedTrainStep(m, *batch(0), steps_per_graph=2).capture(*batch(0))
            for i in range(6):
                gs(*batch(i))
            gs.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(60):
                gs(*batch(i))
            gs.flush()
            torch.cuda.synchronize()
            gt = (time.perf_counter() - t0) / 60
            r.update(graph_ms=gt * 1e3, graph_samples_per_s=B / gt)
        except Exception as exc:  # noqa: BLE001
            r["graph_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
            torch.cuda.synchronize()
        m.model_plan().check_ids()
        res[name] = r
        del m
        torch.cuda.empty_cache()
    except Exception as exc:  # noqa: BLE001
        res[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
print(json.dumps(res, indent=1))
