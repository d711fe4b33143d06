#!/bin/bash
# end of round 4: rocprofv3 --kernel-trace --stats of the bench command (final sources) and the timeline of the sharded
# block step (S steps per hipGraph, direct exchange, one rank)
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
cd /tmp; rm -rf /tmp/prof_d /tmp/prof_s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --warmup-seconds 0 --kernel-iters 2 > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof_d -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof_d -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/timeline.py $t 4 100 > $O/deepfm_graph_timeline.txt 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29556 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 DCTR_SHARDED_EXCHANGE=direct timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o sh -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-parallel --steps 100 --warmup 16 --no-cpu-baseline --no-other-configs --repeats 1 --warmup-seconds 0 --kernel-iters 2 > $O/bench_shard_prof.json 2> $O/bench_shard_prof.err
t=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/timeline.py $t 3 > $O/sharded_block_timeline.txt 2>&1
head -12 $O/deepfm_kernel_stats.csv | cut -c1-150
head -16 $O/sharded_block_timeline.txt; tail -n 2 $O/sharded_block_timeline.txt | cut -c1-300
