#!/bin/bash
# round 4: FiBiNET's bilinear backward re-dealt by owning field -- parity, kernel budget, step time
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_fib8
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_full_golden.py -q --tb=short -k "bilinear or fibinet or Bilinear or FiBiNET" 2>&1 | tail -12) > $O/pytest.log
(timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference_matrix.py -q --tb=short -k "fibinet or FiBiNET" 2>&1 | tail -6) >> $O/pytest.log
cd /tmp; rm -rf /tmp/prof_f
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_f -o fib -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py FiBiNET > $O/prof.log 2>&1
t=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/step_profile.py $t 0 3 > $O/fibinet_step_kernel_budget.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python - > $O/fibinet_bench.json 2> $O/fibinet_bench.err <<'PY'
import sys, json
sys.argv=["bench.py"]
sys.path.insert(0, ".")
import bench, torch
a = bench.parse()
a.steps_per_graph = bench.auto_steps_per_graph(a.steps)
X, y = bench.synth(a, "cuda:0", 0)
print(json.dumps(bench.other_config("fibinet", a, "cuda:0", X, y)))
PY
head -14 $O/fibinet_step_kernel_budget.txt; python -c "
import json; d=json.loads(open('$O/fibinet_bench.json').read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k in ('ms_per_step','value','error')})"
grep -n "passed\|failed" $O/pytest.log
