#!/bin/bash
# queue-based replay: lazy tests, then the default_kwargs leg at several queue depths
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_q
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_checkpoint.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for Q in 0 2 4 8 16; do
  DCTR_LAZY_QROWS=$Q timeout 600 python tools/bench_leg.py default_kwargs > $O/leg_q$Q.json 2> $O/leg.err
  python - $Q <<'PY'
import json,os,sys
q=sys.argv[1]
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_q/leg_q%s.json'%q) if l.startswith('{')][-1])
print('q_rows', q, d.get('error'), round(d.get('ms_per_step',-1),4), round(d.get('steady_state',{}).get('ms_per_step',-1),4), d.get('steady_state',{}).get('error'))
PY
done
DCTR_LAZY_QUEUE=0 timeout 600 python tools/bench_leg.py default_kwargs > $O/leg_noq.json 2> $O/leg.err
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_q/leg_noq.json') if l.startswith('{')][-1])
print('no queue', round(d.get('ms_per_step',-1),4), round(d.get('steady_state',{}).get('ms_per_step',-1),4))
PY
