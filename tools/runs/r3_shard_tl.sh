#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_shard_tl
mkdir -p $O
cd /tmp; rm -rf /tmp/prof_s
MASTER_ADDR=127.0.0.1 MASTER_PORT=29556 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o sh -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-parallel --steps 64 --warmup 16 --no-cpu-baseline --no-other-configs --repeats 1 --warmup-seconds 0 --kernel-iters 2 > $O/bench.json 2> $O/bench.err
t=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/timeline.py $t 4 > $O/timeline.txt 2>&1
