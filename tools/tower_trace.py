#!/usr/bin/env python
"""Decode the tower kernels' phase stamps saved by `bench.py --diag-trace f.npy` (two-layer towers; wall_clock64 ticks
of 10 ns).  Region 0: forward part of k_mlp_train, 1: its backward part, 2: k_mlp_wgrad."""
import json
import sys

import numpy as np

t = np.load(sys.argv[1]).astype(np.int64)


def st(v):
    v = np.asarray(v, dtype=np.float64) * 0.01
    return {"mean": round(float(v.mean()), 2), "p50": round(float(np.percentile(v, 50)), 2),
            "p90": round(float(np.percentile(v, 90)), 2), "max": round(float(v.max()), 2)}


res = {}
f = t[0][t[0][:, 0] > 0]
b = t[1][t[1][:, 0] > 0]
if len(f) and len(b) == len(f):
    t0 = f[:, 0].min()
    res["train"] = {
        "n_wg": int(len(f)), "start_spread": st(f[:, 0] - t0), "fwd_stage_x": st(f[:, 1] - f[:, 0]),
        # (round 4, k_embed_tower_train: descriptors + X tile landed in LDS = the gather's first round trip)
        "fwd_gather_round_trip_1": st(f[:, 10] - f[:, 0]) if f[:, 10].max() > 0 else None,
        "fwd_layer0_mfma": st(f[:, 2] - f[:, 1]), "fwd_layer0_epilogue": st(f[:, 3] - f[:, 2]),
        "fwd_layer0_epilogue:stores": st(f[:, 9] - f[:, 2]) if f[:, 9].max() > 0 else None,
        "fwd_layer0_barrier": st(f[:, 4] - f[:, 3]), "fwd_layer1_mfma": st(f[:, 5] - f[:, 4]),
        "fwd_layer1_epilogue": st(f[:, 6] - f[:, 5]), "fwd_layer1_barrier": st(f[:, 7] - f[:, 6]),
        "fwd_projection": st(f[:, 15] - f[:, 7]), "fwd_total": st(f[:, 15] - f[:, 0]), "head": st(b[:, 0] - f[:, 15]),
        "bwd_stage_top": st(b[:, 1] - b[:, 0]), "bwd_layer1_mfma+epi": st(b[:, 2] - b[:, 1]),
        "bwd_layer1_barrier": st(b[:, 3] - b[:, 2]), "bwd_layer0_mfma+epi": st(b[:, 4] - b[:, 3]),
        "bwd_layer0_barrier": st(b[:, 5] - b[:, 4]), "bwd_total": st(b[:, 15] - b[:, 0]),
        "wg_total": st(b[:, 15] - f[:, 0]), "end": st(b[:, 15] - t0)}
w = t[2][t[2][:, 0] > 0]
if len(w):
    t0 = w[:, 0].min()
    wk = w[w[:, 1] > 0]
    res["wgrad"] = {"n_wg": int(len(w)), "n_gemm_wg": int(len(wk)), "start": st(wk[:, 0] - t0),
                    "mainloop": st(wk[:, 1] - wk[:, 0]), "lds_park+barrier": st(wk[:, 2] - wk[:, 1]),
                    "combine+store": st(wk[:, 3] - wk[:, 2]), "wg_total": st(wk[:, 3] - wk[:, 0]),
                    "end": st(wk[:, 3] - t0),
                    "aux_wg_total (projection blocks: kernel-entry stamp only)": int(len(w) - len(wk))}
print(json.dumps(res, indent=1))
