#!/bin/bash
# entries ordered by gap: lazy tests, then the default_kwargs leg with and without the ordering
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_r
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_checkpoint.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for ORD in 1 0; do
  DCTR_LAZY_ORDER=$ORD timeout 600 python tools/bench_leg.py default_kwargs > $O/leg_ord$ORD.json 2> $O/leg.err
  python - $ORD <<'PY'
import json,os,sys
q=sys.argv[1]
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_r/leg_ord%s.json'%q) if l.startswith('{')][-1])
print('ordered', q, d.get('error'), round(d.get('ms_per_step',-1),4), round(d.get('steady_state',{}).get('ms_per_step',-1),4), d.get('steady_state',{}).get('error'))
PY
done
