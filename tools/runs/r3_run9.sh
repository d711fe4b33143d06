#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_9
mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 0xffffffff 0x3ff0 0xfff0; do
DCTR_MLP_WMASK=$m timeout 300 python tools/tower_bench.py --diag --nx 1 --iters 100 > $O/tower_$m.json 2> $O/tower_$m.err
done
timeout 300 python tools/tower_bench.py --nx 1 --iters 100 > $O/tower_nodiag.json 2> $O/tower_nodiag.err
