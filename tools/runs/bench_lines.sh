#!/bin/bash
# the two bench lines kept under profiles/: the driver's flags and the default flags
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out
( timeout 600 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_driver_flags.err | grep '^{' > $O/bench_driver_flags.json
( timeout 900 python bench.py ) 2> $O/bench.err | grep '^{' > $O/bench.json
python - <<PY
import json
for f in ("bench_driver_flags", "bench"):
    d = json.load(open("$O/" + f + ".json")); r = d["roofline"]
    print(f, round(d["ms_per_step"], 5), round(d["value"]), "frac", round(r["frac"], 4), "traffic", r.get("traffic"), "dom", r.get("dominant_avg_us"), r.get("dominant_frac"),
          {k: round(v.get("ms_per_step", -1), 4) for k, v in d.get("other_configs", {}).items()})
PY
