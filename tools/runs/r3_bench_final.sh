#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_bench_final
mkdir -p $O
( timeout 900 python bench.py ) 2> $O/bench.err | grep '^{' > $O/bench_default_flags.json
( timeout 600 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_drv.err | grep '^{' > $O/bench_driver_flags.json
