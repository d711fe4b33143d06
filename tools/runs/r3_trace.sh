#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_trace
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/tower_bench.py --diag --nx 1 --iters 100 > $O/tower_diag.json 2> $O/tower_diag.err
timeout 300 python tools/tower_bench.py --nx 1 --iters 100 > $O/tower.json 2> $O/tower.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --diag-trace $O/trace.npy > $O/bench_diag.json 2> $O/bench_diag.err
python tools/tower_trace.py $O/trace.npy > $O/trace.json
