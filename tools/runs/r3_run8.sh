#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_8
mkdir -p $O
cd $GRAFT_REPO_ROOT
for nx in 1 8; do
timeout 300 python tools/tower_bench.py --nx $nx --iters 100 > $O/tower_nx$nx.json 2> $O/tower_nx$nx.err
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/tools/tower_bench.py --nx 1 --iters 200) > $O/prof.json 2> $O/prof.err
f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_nx1.csv
