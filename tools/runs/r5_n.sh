#!/bin/bash
# default_kwargs leg (l2 = 1e-5 on tables and linear, adam -> lazy update): per-kernel stats
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_n
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/bench_leg.py default_kwargs --steps 100 > $O/run.log 2>&1
S=$(find $O/prof -name '*kernel_stats.csv' | head -1)
cp $S $O/default_kwargs_kernel_stats.csv
rm -rf $O/prof
head -30 $O/default_kwargs_kernel_stats.csv | cut -c1-200
tail -2 $O/run.log | cut -c1-600
