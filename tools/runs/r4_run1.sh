#!/bin/bash
# round 4, GPU call 1: the fused gather + step engine -- parity first, then A/B timings and a kernel timeline
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_1
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_step_engine.py -x -q 2>&1 | tail -25) > $O/pytest_engine.log
(timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py -x -q 2>&1 | tail -12) > $O/pytest_mlp_deepfm.log
B="--no-other-configs --no-cpu-baseline --steps 200 --warmup 20"
timeout 300 python bench.py $B > $O/bench_engine.json 2> $O/bench_engine.err
DCTR_STEP_ENGINE=0 timeout 300 python bench.py $B > $O/bench_old.json 2> $O/bench_old.err
DCTR_STEP_TOPOLOGY=fused_flags timeout 300 python bench.py $B > $O/bench_flags.json 2> $O/bench_flags.err
DCTR_STEP_TOPOLOGY=serial timeout 300 python bench.py $B > $O/bench_serial.json 2> $O/bench_serial.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 6 > $O/timeline.txt
(timeout 1200 python -m pytest tests/test_gpu_full_golden.py tests/test_gpu_step_topology.py tests/test_gpu_fit.py -x -q 2>&1 | tail -12) > $O/pytest_golden.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "M/s", round(d["value"]/1e6,2), "upd in-step us", round(r.get("avg_us",0),2), "frac", round(r.get("frac",0),3), "loss", d.get("final_loss"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -3 $O/pytest_*.log
