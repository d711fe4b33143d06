#!/bin/bash
# default-kwargs leg against the sweep's K (windows per table): gpurun -- bash tools/runs/sweep_k.sh [K...]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_k; mkdir -p $O
for k in ${@:-32 16 24 48 32 16}; do
  DCTR_LAZY_SWEEP_K=$k timeout 600 python tools/bench_leg.py default_kwargs --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('K=$k', d.get('ms_per_step'), (d.get('steady_state') or {}).get('ms_per_step'))" | tee -a $O/summary.txt
done
