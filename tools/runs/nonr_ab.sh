#!/bin/bash
# A/B on one box: the replay loop with / without its Newton corrections (gpurun_in_nonr.so built with -DDCTR_LAZY_REPLAY_NR=0)
export TMPDIR=/tmp
export DCTR_LAZY_SWEEP_PERSIST=0
O=$GRAFT_REPO_ROOT/gpurun_out/nonr_ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
L=deepctr-torch_amd/deepctr_torch/_hip/libdctr_hip.so
cp $L /tmp/ship.so
for rep in 1 2; do
  for v in ship nonr; do
    if [ $v = ship ]; then cp /tmp/ship.so $L; else cp gpurun_in_nonr.so $L; fi
    timeout 600 python tools/bench_leg.py default_kwargs --steps 20 --warmup 5 > $O/leg_${v}_$rep.json 2> $O/leg_${v}_$rep.err
    python - <<PY | tee -a $O/summary.txt
import json
def last(p):
    try: return json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e: return {"error": str(e)}
d=last("$O/leg_${v}_$rep.json")
print("$v $rep default_kwargs", d.get("ms_per_step"), (d.get("steady_state") or {}).get("ms_per_step"), d.get("error"))
PY
  done
done
cp gpurun_in_nonr.so $L
timeout 1200 python -m pytest tests -x -q -m gpu -k "lazy or default or scratch or adam or l2 or reg" 2>&1 | tail -8 | tee $O/tests_nonr.log
cp /tmp/ship.so $L
