#!/bin/bash
# A/B on one box: the shipped library (chain launches at raised issue priority) against a build with -DDCTR_NO_STEP_PRIORITY
# (gpurun_in_noprio.so at the repo root): gpurun -- bash tools/runs/prio_ab.sh
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prio_ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
L=deepctr-torch_amd/deepctr_torch/_hip/libdctr_hip.so
cp $L /tmp/prio.so
for rep in 1 2; do
  for v in prio noprio; do
    if [ $v = prio ]; then cp /tmp/prio.so $L; else cp gpurun_in_noprio.so $L; fi
    DCTR_LAZY_SWEEP_WAVES=0 timeout 600 python tools/bench_leg.py default_kwargs --steps 20 --warmup 5 > $O/leg_${v}_$rep.json 2> $O/leg_${v}_$rep.err
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $O/head_${v}_$rep.json 2> $O/head_${v}_$rep.err
    python - <<PY | tee -a $O/summary.txt
import json
def last(p):
    try: return json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e: return {"error": str(e)}
d=last("$O/leg_${v}_$rep.json"); h=last("$O/head_${v}_$rep.json")
print("$v $rep default_kwargs", d.get("ms_per_step"), (d.get("steady_state") or {}).get("ms_per_step"), "headline", h.get("ms_per_step"))
PY
  done
done
cp /tmp/prio.so $L
