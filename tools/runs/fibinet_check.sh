#!/bin/bash
# FiBiNET after a change: its tests, then the leg with / without the aligned slab
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/fibinet_check; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "fibinet or FiBiNET or bilinear or pairwise or senet" 2>&1 | tail -3 | tee $O/tests.log
bash tools/runs/leg.sh fibinet
cp gpurun_out/leg_fibinet/kernels.txt $O/kernels_aligned.txt
DCTR_SLAB_ALIGN=0 timeout 600 python tools/bench_leg.py fibinet --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fibinet DCTR_SLAB_ALIGN=0', d.get('ms_per_step'))"
timeout 600 python tools/bench_leg.py fibinet --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fibinet aligned', d.get('ms_per_step'))"
