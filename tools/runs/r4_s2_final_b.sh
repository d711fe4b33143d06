#!/bin/bash
# end of round 4: PMC traffic of the update kernel on the final sources (two --pmc passes, --kernel-trace only), then the
# bench line with the driver's default flags
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
cd $GRAFT_REPO_ROOT
PMC_OPTS=adagrad PMC_BATCHES=4096 bash tools/pmc_traffic.sh > $O/pmc.log 2>&1
tail -n 4 $O/pmc.log | cut -c1-200
python - <<'PY'
import json, shutil
d = json.load(open("gpurun_out/pmc_summary.json"))
ok = any(k.startswith("embed_update_adagrad") for k in d.get("kernels", {}))
print("pmc summary kernels:", [k for k in d.get("kernels", {})][:6], "usable:", ok)
if ok:
    shutil.copy("gpurun_out/pmc_summary.json", "profiles/r04_pmc_traffic.json")
PY
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
s=d["hot_path"]["saturating"]
print("saturating", s["update_path_frac_of_hbm_peak"], s["gather_frac_of_hbm_peak"])
print({k:(v.get("ms_per_step"), v.get("value")) for k,v in d.get("other_configs",{}).items()})
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","kind","cores")})
PY
