#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_spg
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for s in 8 10 16 20 25 32 40; do
  timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 800 --warmup 20 --repeats 3 --steps-per-graph $s 2>/dev/null | grep '^{' > $O/s${s}_$rep.json
done
done
for f in $O/*.json; do python -c "import json,sys;d=json.load(open('$f'));print('$f'.split('/')[-1], round(d['ms_per_step'],5), d['config'].get('steps_per_graph'))"; done > $O/summary.txt
