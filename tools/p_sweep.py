#!/usr/bin/env python
"""DeepFM step time against the update's partitions per unit P (diag library: dctr_dbg_update_trace(NULL, P) overrides
pick_p for the pre-pass and the update alike).  The default P = B / 96 gives 1118 workgroups at B = 4096; beside a resident
k_mlp_wgrad wave (124 VGPRs) a SIMD holds 4 update waves, i.e. 4096 on the chip: fewer, larger partitions fit one residency
round.     python tools/p_sweep.py [P ...]     one JSON line per P"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch._hip import lib as L  # noqa: E402

L.use_diag_library()
import bench  # noqa: E402

Ps = [int(a) for a in sys.argv[1:]] or [0, 32, 36, 39, 48]
sys.argv = ["bench.py"]
args = bench.parse()
dev = "cuda:0"
torch.cuda.set_device(0)
X, y = bench.synth(args, dev, 0)
for P in Ps:
    L.lib().dctr_dbg_update_trace(None, P if P > 0 else -1)
    model = bench.build_model(args, dev)
    S = bench.auto_steps_per_graph(200)
    elapsed, out, graphed, did, r, times = bench.time_steps(model, X, y, args.batch, 200, 20, S, True, 5, 1.0)
    model.model_plan().check_ids()
    upd = bench.time_update_in_step(model, X, y, args.batch)
    print(json.dumps({"P": P, "ms_per_step": elapsed / 200 * 1e3, "all": [round(t / 200 * 1e3, 5) for t in times],
                      "graphed": graphed, "update_in_step_us": upd["avg_us"] if upd else None,
                      "loss": float(out[0].item())}), flush=True)
    del model, r
    torch.cuda.empty_cache()
