#!/bin/bash
# decomposition of k_prepass_sort at B = 262 144 (diag build: DCTR_PREPASS_DBG 1 no stores, 2 no rank sort, 4 no walk)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_4
mkdir -p $O
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 7; do
echo "dbg=$d"; DCTR_PREPASS_DBG=$d timeout 120 python tools/prepass_bench.py 262144 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['B'], d['two_level_us'], d['keys_equal'])"
done
