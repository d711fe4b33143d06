#!/bin/bash
# timing experiments: which graph edges cost what (DCTR_DBG_EDGES drops edges: results are wrong, timelines are not)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/edges
mkdir -p $O
cd $GRAFT_REPO_ROOT
for e in ${@:-1 2 3 7}; do
  rm -rf $O/trace
  DCTR_DBG_EDGES=$e DCTR_STEP_WAIT_US=20000 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating --warmup-seconds 0.2 --repeats 1 --kernel-iters 2 > $O/trace_$e.log 2>&1
  f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  echo "=== DCTR_DBG_EDGES=$e"
  python tools/timeline.py $f 2 90 > $O/timeline_$e.txt 2>&1
  tail -14 $O/timeline_$e.txt
done
rm -rf $O/trace
