#!/usr/bin/env python
"""Device-side hop latencies of the "flags" step topology: runs the DeepFM bench model under DCTR_STEP_TOPOLOGY=flags in
hipGraph replay (multi-step graphs) and reads the sync block's time stamps (s_memrealtime, 10 ns) after each group of
replays -- they belong to the last step of the last replay.
    python tools/step_hops.py > gpurun_out/step_hops.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
os.environ["DCTR_STEP_TOPOLOGY"] = "flags"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

B = 4096


class A:
    vocab, batch, optimizer, ids = 1_000_000, B, "adagrad", "uniform"


dev = "cuda:0"
model = bench.build_model(A, dev)
X, y = bench.synth(A, dev, 0)
for i in range(3):
    model._train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
from deepctr_torch._hip.graph import GraphedTrainStep  # noqa: E402
S = 8
g = GraphedTrainStep(model, X[:B], y[:B], steps_per_graph=S, inputs_ready=True).capture(X[:B], y[:B])
slab = model._fused["slab"]
rows = []
for rep in range(60):
    for k in range(3):          # 3 replays of 8 steps back to back; stamps of the very last step
        j = ((rep * 3 + k) * S) % 56
        g.step_block(X[j * B:(j + S) * B], y[j * B:(j + S) * B])
    torch.cuda.synchronize()
    rows.append(slab._sync.cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
model.model_plan().check_ids()
r = np.array(rows[5:])
t_sig, g_sig = r[:, 3], r[:, 7]
tw0, tw1, gw0, gw1 = r[:, 16], r[:, 17], r[:, 18], r[:, 19]


def us(a):
    a = ((a + (1 << 31)) % (1 << 32)) - (1 << 31)
    return {"median": round(float(np.median(a)) * 0.01, 2), "p10": round(float(np.percentile(a, 10)) * 0.01, 2),
            "p90": round(float(np.percentile(a, 90)) * 0.01, 2)}


print(json.dumps({
    "gather_signal -> tower_waiter_end": us(gw1 - g_sig),
    "tower_waiter: start -> end": us(gw1 - gw0),
    "tower_waiter_end -> tower_signal (tower kernel + its dispatch)": us(t_sig - gw1),
    "tower_signal -> update_waiter_end": us(tw1 - t_sig),
    "update_waiter: start -> end": us(tw1 - tw0),
    "gather_signal -> tower_signal": us(t_sig - g_sig),
    "gather_signal -> update_waiter_end": us(tw1 - g_sig),
}, indent=1))
