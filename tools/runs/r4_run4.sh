#!/bin/bash
# round 4, GPU call 4: full-size goldens (incl. the new Zipf / VarLen fixtures), Zipf bench run, fit_api leg
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_4
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_full_golden.py -q --tb=short 2>&1 | tail -30) > $O/pytest_golden.log
cp gpurun_out/full_golden_errors.json $O/ 2>/dev/null
timeout 300 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --ids zipf > $O/bench_zipf.json 2> $O/bench_zipf.err
timeout 300 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_uniform.json 2> $O/bench_uniform.err
timeout 300 python - > $O/fit_api.json 2> $O/fit_api.err <<'PY'
import sys, json
sys.argv=["bench.py"]
sys.path.insert(0, ".")
import bench, torch
a = bench.parse()
X, y = bench.synth(a, "cuda:0", 0)
print(json.dumps(bench.fit_api(a, "cuda:0", X, y)))
PY
tail -n 4 $O/pytest_golden.log
python - <<'PY'
import json
for t in ("zipf","uniform"):
    d=json.loads(open("gpurun_out/r4_4/bench_%s.json"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["value"], d["roofline"]["avg_us"], d["roofline"]["standalone_avg_us"])
print(open("gpurun_out/r4_4/fit_api.json").read()[:1200])
e=json.load(open("gpurun_out/r4_4/full_golden_errors.json"))
for w in e["worst_by_err_over_bar"][:8]: print(w)
PY
