#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_topo2
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in update_side gather_side update_side gather_side; do
  DCTR_STEP_TOPOLOGY=$t timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 > $O/bench_$t.json 2> $O/bench_$t.err
  python -c "import json;d=json.load(open('$O/bench_$t.json'));print('$t',d['value'],d['ms_per_step'])" >> $O/summary.txt
done
(cd /tmp && DCTR_STEP_TOPOLOGY=gather_side timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 > $O/timeline.txt
