#!/bin/bash
# the headline line only (no other legs, no cpu baseline): gpurun -- bash tools/runs/bench_head.sh [extra bench args]
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bench_head
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline "$@" > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
r=d["roofline"]
print("ms/step", round(d["ms_per_step"],5), "frac", round(r["frac"],4), "avg_us", round(r["avg_us"],2))
for k in ("dominant_kernel","dominant_bound","dominant_avg_us","dominant_frac","update_avg_us_in_graph","update_frac_in_graph","sclk_mhz","power_w"):
    print(" ", k, r.get(k))
print(" in_graph", r.get("in_graph"))
sat=(d.get("hot_path") or {}).get("saturating")
if sat: print(" saturating", {k: round(v["frac_of_hbm_peak"],3) for k,v in sat["kernels"].items()})
PY
