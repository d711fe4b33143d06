#!/bin/bash
# round 4: the step engine WITHOUT hipGraphs -- host pace and the cross-queue dependency latency of plain stream events
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_eager
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-graph --no-other-configs --no-cpu-baseline --steps 400 --warmup 50 --repeats 3 2> $O/err.txt | grep '^{' > $O/bench_eager.json
python -c "
import json; d=json.load(open('$O/bench_eager.json')); print('eager', d['ms_per_step'], d['value'])"
cd /tmp; rm -rf /tmp/prof_e
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -o e -- python $GRAFT_REPO_ROOT/bench.py --no-graph --no-other-configs --no-cpu-baseline --steps 64 --warmup 16 --repeats 1 --warmup-seconds 0 --kernel-iters 2 > $O/prof.json 2> $O/prof.err
t=$(find /tmp/prof_e -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/timeline.py $t 3 > $O/timeline.txt 2>&1
head -24 $O/timeline.txt; tail -1 $O/timeline.txt
