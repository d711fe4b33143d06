#!/bin/bash
# default-kwargs leg + headline head with raised chain priority: gpurun -- bash tools/runs/sweep_prio.sh [waves...]
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sweep_prio
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_step_engine.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
for w in ${@:-0 3 4}; do
  DCTR_LAZY_SWEEP_WAVES=$w timeout 600 python tools/bench_leg.py default_kwargs --steps 20 --warmup 5 > $O/leg_$w.json 2> $O/leg_$w.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open("$O/leg_$w.json") if l.startswith("{")][-1])
    print("waves $w", d.get("ms_per_step"), (d.get("steady_state") or {}).get("ms_per_step"), d.get("final_loss"))
except Exception as e: print("waves $w failed", e); print(open("$O/leg_$w.err").read()[-800:])
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $O/headline.json 2> $O/headline.err
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open("$O/headline.json") if l.startswith("{")][-1])
    print("headline", d.get("ms_per_step"), d.get("value"))
except Exception as e: print("headline failed", e); print(open("$O/headline.err").read()[-800:])
PY
