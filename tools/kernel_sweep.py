#!/usr/bin/env python
"""The two hand-written embedding kernels (gather, fused update) under each table layout (_hip/layout.py) at the bench's
launch (B = 4096), the 8-GPU global batch (32 768) and the saturating launch of SURVEY 8(d) (262 144).
    python tools/kernel_sweep.py [layouts] [optimizers] > gpurun_out/kernel_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = "cuda:0"
layouts = (sys.argv[1] if len(sys.argv) > 1 else "contiguous,interleaved,block").split(",")
opts = (sys.argv[2] if len(sys.argv) > 2 else "adagrad").split(",")
sizes = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "4096,32768,262144").split(",")]


class A:
    vocab, batch, optimizer = 1_000_000, 4096, "adagrad"


res = {}
for layout in layouts:
    os.environ["DCTR_TABLE_LAYOUT"] = layout
    for opt in opts:
        A.optimizer = opt
        model = bench.build_model(A, dev)
        p = model.embedding_dict["C1"].weight
        res.setdefault(layout, {})[opt] = {"row_stride_floats": int(p.stride(0))}
        for Bsz in sizes:
            A.batch = Bsz
            mult = 8 if Bsz <= 32_768 else 2
            gen = torch.Generator().manual_seed(0)
            X = torch.cat([torch.randint(0, A.vocab, (mult * Bsz, 26), generator=gen).float(),
                           torch.rand(mult * Bsz, 13, generator=gen)], 1).to(dev)
            k = bench.time_hot_kernels(model, X, Bsz, 20 if Bsz <= 32_768 else 6, opt, ring=mult)
            alg = bench.algorithmic_bytes(Bsz, opt)
            for name in k:
                k[name]["alg_GBs"] = alg[name] / (k[name]["avg_us"] * 1e-6) / 1e9
            res[layout][opt]["B%d" % Bsz] = {n: {"avg_us": round(v["avg_us"], 1), "min_us": round(v["min_us"], 1),
                                                 "alg_GBs": round(v["alg_GBs"])} for n, v in k.items()}
            del X
        del model
        torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
