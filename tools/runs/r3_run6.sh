#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_6
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 0 1; do
HIP_FORCE_DEV_KERNARG=$v timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_kernarg$v.json 2> $O/bench_kernarg$v.err
(cd /tmp && HIP_FORCE_DEV_KERNARG=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/prof_$v.json 2> $O/prof_$v.err
f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$v.csv
done
