#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_final
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
python - <<'PY'
import json
for t in ("default_flags","driver_flags"):
    d=json.loads(open("gpurun_out/r4_final/bench_%s.json"%t).read().strip().splitlines()[-1])
    print(t, "ms/step", round(d["ms_per_step"],4), "M/s", round(d["value"]/1e6,2), "roof", round(d["roofline"]["frac"],3), d["roofline"]["timed"])
    for k,v in d.get("other_configs",{}).items(): print("  ",k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_step","value","error","vs_step_runner")}, v.get("shuffle_false",{}).get("ms_per_step"), v.get("small_epochs",{}).get("ms_per_step"))
    print("   cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"], round(d["cpu_baseline"]["ms_per_step"],1))
PY
