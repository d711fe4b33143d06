#!/bin/bash
# round 4: symmetric first CIN layer -- parity + xDeepFM step time (sym on / off) + kernel budget
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_cin5
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cin.py -q --tb=short 2>&1 | tail -8) > $O/pytest.log
(timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_full_golden.py tests/test_gpu_reference_matrix.py -q --tb=short -k "xdeepfm or xDeepFM or cin or CIN" 2>&1 | tail -8) >> $O/pytest.log
for sym in 1 0; do
DCTR_CIN_SYM=$sym timeout 600 python - > $O/xdeepfm_sym$sym.json 2> $O/xdeepfm_sym$sym.err <<'PY'
import sys, json
sys.argv=["bench.py"]
sys.path.insert(0, ".")
import bench, torch
a = bench.parse()
a.steps_per_graph = bench.auto_steps_per_graph(a.steps)
X, y = bench.synth(a, "cuda:0", 0)
print(json.dumps(bench.other_config("xdeepfm", a, "cuda:0", X, y)))
PY
python -c "
import json; d=json.loads(open('$O/xdeepfm_sym$sym.json').read().strip().splitlines()[-1]); print('sym=$sym', {k:v for k,v in d.items() if k in ('ms_per_step','value','error')})"
done
cd /tmp; rm -rf /tmp/prof_x
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_x -o x -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py xDeepFM > $O/prof.log 2>&1
t=$(find /tmp/prof_x -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/step_profile.py $t 1 4 > $O/xdeepfm_step_kernel_budget.txt 2>&1
head -12 $O/xdeepfm_step_kernel_budget.txt
grep -n "passed\|failed\|Error" $O/pytest.log
