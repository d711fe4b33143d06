#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_slices
mkdir -p $O
cd $GRAFT_REPO_ROOT
for s in 7 14 7 14; do
  DCTR_WGRAD_SLICES=$s timeout 300 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 --warmup-seconds 0.5 --diag-trace /tmp/t.npy 2>/dev/null | grep '^{' > $O/s$s.json
  python -c "import json;d=json.load(open('$O/s$s.json'));print('S=$s',d['ms_per_step'])" >> $O/summary.txt
done
