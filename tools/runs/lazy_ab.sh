#!/bin/bash
# A/B on one box: the shipped library against a variant .so at the repo root on the default-kwargs leg (+ the lazy tests with
# the shipped one):  gpurun -- bash tools/runs/lazy_ab.sh <variant.so> [reps]
export TMPDIR=/tmp
V=$1; R=${2:-2}
O=$GRAFT_REPO_ROOT/gpurun_out/lazy_ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
L=deepctr-torch_amd/deepctr_torch/_hip/libdctr_hip.so
cp $L /tmp/ship.so
timeout 1200 python -m pytest tests -x -q -m gpu -k "lazy or default or scratch or adam or l2 or reg" 2>&1 | tail -3 | tee $O/tests_ship.log
for rep in $(seq 1 $R); do
  for v in ship variant; do
    if [ $v = ship ]; then cp /tmp/ship.so $L; else cp $V $L; fi
    timeout 600 python tools/bench_leg.py default_kwargs --steps 20 --warmup 5 > $O/leg_${v}_$rep.json 2> $O/leg_${v}_$rep.err
    python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open("$O/leg_${v}_$rep.json") if l.startswith("{")][-1])
    print("$v $rep default_kwargs", d.get("ms_per_step"), (d.get("steady_state") or {}).get("ms_per_step"), d.get("error"))
except Exception as e: print("$v $rep failed", e)
PY
  done
done
cp /tmp/ship.so $L
