#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_flags2
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in update_side flags update_side flags; do
  DCTR_STEP_TOPOLOGY=$t timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 > $O/bench_$t.json 2> $O/bench_$t.err
  python -c "import json;d=json.load(open('$O/bench_$t.json'));print('$t',d['value'],d['ms_per_step'])" >> $O/summary.txt
done
timeout 300 python tools/step_hops.py > $O/hops.json 2> $O/hops.err
timeout 600 python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -3 > $O/pytest.txt
