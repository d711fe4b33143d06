#!/bin/bash
# round 4, GPU call 3: pre-pass relocated behind the update; block layout A/B; fit() host profile
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_3
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_step_engine.py -q --tb=short 2>&1 | tail -30) > $O/pytest_engine.log
B="--no-other-configs --no-cpu-baseline --steps 200 --warmup 20"
timeout 300 python bench.py $B > $O/bench_engine.json 2> $O/bench_engine.err
DCTR_TABLE_LAYOUT=block timeout 300 python bench.py $B > $O/bench_block.json 2> $O/bench_block.err
DCTR_TABLE_LAYOUT=contiguous timeout 300 python bench.py $B > $O/bench_contig.json 2> $O/bench_contig.err
timeout 300 python tools/fit_profile.py > $O/fit_profile.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 4 100 > $O/timeline.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "M/s", round(d["value"]/1e6,2), "upd in-step us", round(r.get("avg_us",0),2), "frac", round(r.get("frac",0),3))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -n 3 $O/pytest_engine.log
head -5 $O/fit_profile.txt
