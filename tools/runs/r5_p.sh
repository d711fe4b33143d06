#!/bin/bash
# default_kwargs leg with its steady-state measurement (1024 distinct batches)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_p
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/bench_leg.py default_kwargs > $O/leg.json 2> $O/leg.err
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_p/leg.json') if l.startswith('{')][-1])
print(d.get('error'), d.get('ms_per_step'), d.get('rows_repeat_every_steps'), d.get('steady_state'))
PY
tail -3 $O/leg.err
