#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/tower_bench.py --diag --trace --nx 1 --iters 20 > $O/tower_nx1.json 2> $O/tower.err
timeout 300 python tools/tower_bench.py --diag --trace --nx 8 --iters 20 > $O/tower_nx8.json 2>> $O/tower.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 1 --diag-trace $O/trace_base.npy > $O/bench_base.json 2> $O/bench_base.err
python tools/tower_trace.py $O/trace_base.npy > $O/trace_base.json
