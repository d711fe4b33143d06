#!/bin/bash
# the default-kwargs leg with the sweep as a persistent launch (1) / a plain grid (0), alternating on one box: gpurun -- bash tools/runs/sweep_waves.sh [values...]
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sweep_waves
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lazy.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
n=0
for w in ${@:-1 0 1 0}; do
  n=$((n+1))
  DCTR_LAZY_SWEEP_PERSIST=$w timeout 600 python tools/bench_leg.py default_kwargs --steps 20 --warmup 5 > $O/leg_${w}_$n.json 2> $O/leg_${w}_$n.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open("$O/leg_${w}_$n.json") if l.startswith("{")][-1])
    print("persist $w", d.get("ms_per_step"), (d.get("steady_state") or {}).get("ms_per_step"), d.get("final_loss"))
except Exception as e: print("waves $w failed", e); print(open("$O/leg_${w}_$n.err").read()[-800:])
PY
done
