#!/bin/bash
# the fused pairs + first-layer backward (csrc/bilinear_wide.hip): its tests, then the FiBiNET leg with and without it
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/wide_check; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bilinear_wide.py -x -q 2>&1 | tail -15 | tee $O/tests_wide.log
if [ "$1" != "quick" ]; then
timeout 1500 python -m pytest tests -x -q -m gpu -k "fibinet or FiBiNET or bilinear or pairwise or senet" 2>&1 | tail -5 | tee $O/tests.log
fi
bash tools/runs/leg.sh fibinet
cp gpurun_out/leg_fibinet/kernels.txt $O/kernels_wide.txt
DCTR_BILINEAR_WIDE=0 timeout 600 python tools/bench_leg.py fibinet --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fibinet DCTR_BILINEAR_WIDE=0', d.get('ms_per_step'))" | tee $O/ab.txt
timeout 600 python tools/bench_leg.py fibinet --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fibinet fused', d.get('ms_per_step'))" | tee -a $O/ab.txt
