#!/bin/bash
# chained step (two linear graphs): headline without the profiler, a few settings
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/chain_quick
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" DCTR_DBG_IGNORE_WAIT=1 timeout 300 python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating --kernel-iters 2 > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.json") if l.startswith("{")][-1])
    print("$tag", round(d["ms_per_step"],5), d["timing"]["ms_per_step_all"], d["final_loss"])
except Exception as e:
    print("$tag failed", e)
print(open("$O/$tag.err").read()[-600:])
PY
}
STEPS=20 run chain20 DCTR_STEP_TOPOLOGY=weights_flag
STEPS=100 run chain100 DCTR_STEP_TOPOLOGY=weights_flag
STEPS=20 run old20 DCTR_STEP_TOPOLOGY=update_side
STEPS=100 run old100 DCTR_STEP_TOPOLOGY=update_side
