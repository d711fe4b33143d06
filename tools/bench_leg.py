#!/usr/bin/env python
"""One leg of bench.py's other_configs (a DeepFM variant, or xdeepfm / fibinet) on its own (for rocprofv3: the process then holds that leg's kernels only).
    python tools/bench_leg.py deepfm_varlen [--steps 100]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
leg = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench  # noqa: E402

args = bench.parse()
if args.steps_per_graph <= 0:
    args.steps_per_graph = bench.auto_steps_per_graph(args.steps)
import torch  # noqa: E402
torch.cuda.set_device(0)
X, y = bench.synth(args, "cuda:0", 0)
if leg in bench.OTHER:
    print(json.dumps(bench.other_config(leg, args, "cuda:0", X, y)))
else:
    print(json.dumps(bench.deepfm_leg(leg, args, "cuda:0", X, y)))
