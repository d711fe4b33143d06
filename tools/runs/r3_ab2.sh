#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_ab2
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  (cd _ab_old && timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 3) 2>/dev/null | grep '^{' > $O/old_$i.json
  timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 2>/dev/null | grep '^{' > $O/new_$i.json
  timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 --steps-per-graph 25 2>/dev/null | grep '^{' > $O/new25_$i.json
done
for f in $O/*.json; do python -c "import json,sys;d=json.load(open('$f'));print('$f'.split('/')[-1], round(d['ms_per_step'],5), d['config'].get('steps_per_graph'))"; done > $O/summary.txt
