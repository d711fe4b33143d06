#!/bin/bash
# A/B on one box: the shipped library against a variant .so at the repo root, alternating:
#   gpurun -- bash tools/runs/lib_ab.sh <variant.so> <reps> [bench args...]   (headline only, 100-step blocks)
export TMPDIR=/tmp
V=$1; R=${2:-3}; shift; shift
O=$GRAFT_REPO_ROOT/gpurun_out/lib_ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
L=deepctr-torch_amd/deepctr_torch/_hip/libdctr_hip.so
cp $L /tmp/ship.so
for rep in $(seq 1 $R); do
  for v in ship variant; do
    if [ $v = ship ]; then cp /tmp/ship.so $L; else cp $V $L; fi
    timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline "$@" > $O/head_${v}_$rep.json 2> $O/head_${v}_$rep.err
    python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open("$O/head_${v}_$rep.json") if l.startswith("{")][-1])
    r=d["roofline"]; print("$v $rep ms/step", round(d["ms_per_step"],5), "upd_us", r.get("update_avg_us_in_graph"), "tower_us", r.get("dominant_avg_us"), "sclk", r.get("sclk_mhz"))
except Exception as e: print("$v $rep failed", e)
PY
  done
done
cp /tmp/ship.so $L
