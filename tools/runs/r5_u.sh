#!/bin/bash
# per-step sweep: lazy tests, then the default_kwargs leg for K = 256 / 128 / 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_u
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_checkpoint.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for K in 256 128 0; do
  DCTR_LAZY_SWEEP_K=$K timeout 600 python tools/bench_leg.py default_kwargs > $O/leg_k$K.json 2> $O/leg.err
  python - $K <<'PY'
import json,os,sys
q=sys.argv[1]
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_u/leg_k%s.json'%q) if l.startswith('{')][-1])
print('sweep K', q, d.get('error'), round(d.get('ms_per_step',-1),4), round(d.get('steady_state',{}).get('ms_per_step',-1),4), d.get('steady_state',{}).get('error'))
PY
done
