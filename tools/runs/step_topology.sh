#!/bin/bash
# DeepFM step engine: parity tests of the weights_flag topology, the headline under both topologies, a kernel timeline.
#   gpurun -- bash tools/runs/step_topology.sh [tests|bench|trace ...]   (default: all three)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/step_topology
mkdir -p $O
cd $GRAFT_REPO_ROOT
WHAT="${@:-tests bench trace}"
for w in $WHAT; do
case $w in
tests)
  timeout 900 python -m pytest tests/test_gpu_step_engine.py tests/test_gpu_step_topology.py tests/test_gpu_deepfm.py tests/test_gpu_mlp.py -q -m gpu -x --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
  tail -12 $O/pytest.txt ;;
bench)
  for topo in weights_flag update_side; do
    DCTR_STEP_TOPOLOGY=$topo timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating > $O/bench_$topo.json 2> $O/bench_$topo.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$topo.json") if l.startswith("{")][-1])
    print("$topo", round(d["ms_per_step"],5), d["final_loss"], d["roofline"]["avg_us"], (d["roofline"].get("dominant") or {}).get("avg_us"))
except Exception as e:
    print("$topo failed", e); print(open("$O/bench_$topo.err").read()[-1500:])
PY
  done ;;
trace)
  rm -rf $O/trace
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating --warmup-seconds 0.2 --repeats 1 > $O/trace.log 2>&1
  f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  cp $f $O/kernel_trace.csv
  python tools/timeline.py $f 24 90 > $O/timeline.txt 2>&1
  tail -150 $O/timeline.txt | head -150
  s=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $s $O/kernel_stats.csv; head -12 $O/kernel_stats.csv
  rm -rf $O/trace ;;
esac
done
