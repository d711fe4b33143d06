#!/bin/bash
# round 4, second session: two-level pre-pass of large batches (A/B + bit-equality), update tests, P sweep of the step
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_2
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/prepass_bench.py > $O/prepass_uniform.jsonl 2> $O/prepass_uniform.err
timeout 300 python tools/prepass_bench.py --zipf > $O/prepass_zipf.jsonl 2> $O/prepass_zipf.err
(timeout 600 python -m pytest tests/test_gpu_update.py -q --tb=short -x 2>&1 | tail -8) > $O/pytest.log

cat $O/prepass_uniform.jsonl $O/prepass_zipf.jsonl | cut -c1-420
tail -n 3 $O/prepass_uniform.err $O/prepass_zipf.err

tail -4 $O/pytest.log
timeout 300 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_s2_2/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(json.dumps(d["hot_path"]["saturating"]))
PY
