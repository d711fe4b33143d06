#!/bin/bash
# round 4, GPU call 2: parity of the step engine after pinning the roundings; graph-region timeline
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_2
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_step_engine.py -q --tb=short -k "not fused_flags" 2>&1 | tail -40) > $O/pytest_engine.log
(timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_fit.py -q --tb=short 2>&1 | tail -15) > $O/pytest_mlp_deepfm.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 4 100 > $O/timeline.txt
(cd /tmp && DCTR_STEP_TOPOLOGY=fused_flags timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2 -o flags -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 48 --warmup 4 --kernel-iters 2 --repeats 1 --warmup-seconds 0) > $O/bench_flags.json 2> $O/bench_flags.err
t=$(find /tmp/prof2 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 3 100 > $O/timeline_flags.txt
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_full.json 2> $O/bench_full.err
tail -4 $O/pytest_engine.log $O/pytest_mlp_deepfm.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_2/bench_full.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"])
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("traffic_detail","in_step")})
print("sat", json.dumps(d["hot_path"].get("saturating"))[:900])
for k,v in d.get("other_configs",{}).items(): print(k, {a:b for a,b in v.items() if a in ("ms_per_step","value","error","vs_step_runner")})
print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k!="variants"})
PY
