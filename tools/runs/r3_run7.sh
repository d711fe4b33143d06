#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_7
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --diag-trace $O/trace.npy > $O/bench_diag.json 2> $O/bench_diag.err
python tools/tower_trace.py $O/trace.npy > $O/trace.json
