#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_3
mkdir -p $O
cd $GRAFT_REPO_ROOT
for topo in update_side tower_side gather_side; do
DCTR_STEP_TOPOLOGY=$topo timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_$topo.json 2> $O/bench_$topo.err
done
for topo in tower_side gather_side; do
(cd /tmp && DCTR_STEP_TOPOLOGY=$topo timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$topo -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/prof_$topo.json 2> $O/prof_$topo.err
t=$(find /tmp/prof_$topo -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 > $O/timeline_$topo.txt
done
