#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_ac
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_checkpoint.py tests/test_gpu_fit.py tests/test_gpu_reference_matrix.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
run() {
  env "$@" timeout 600 python tools/bench_leg.py default_kwargs > $O/leg.json 2> $O/leg.err
  python - "$*" <<'PY'
import json,os,sys
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_ac/leg.json') if l.startswith('{')][-1])
print(sys.argv[1], d.get('error'), round(d.get('ms_per_step',-1),4), round(d.get('steady_state',{}).get('ms_per_step',-1),4), d.get('steady_state',{}).get('error'))
PY
}
run DCTR_LAZY_SWEEP_ASYNC=1
run DCTR_LAZY_SWEEP_ASYNC=0
