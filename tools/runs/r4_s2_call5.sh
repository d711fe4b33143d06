#!/bin/bash
# two-level pre-pass, final constants: A/B, update + sharded + step-engine tests, bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_5
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/prepass_bench.py > $O/prepass_uniform.jsonl 2>/dev/null
timeout 300 python tools/prepass_bench.py --zipf > $O/prepass_zipf.jsonl 2>/dev/null
cat $O/prepass_uniform.jsonl $O/prepass_zipf.jsonl | cut -c1-260
(timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_step_engine.py tests/test_gpu_fullsize.py tests/test_gpu_direct_exchange.py -q --tb=short -x 2>&1 | tail -5) | tee $O/pytest.log
timeout 300 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_s2_5/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
s=d["hot_path"]["saturating"]
print({k:(round(v["avg_us"],1), round(v["frac_of_hbm_peak"],3)) for k,v in s["kernels"].items()}, s["update_path_frac_of_hbm_peak"], s["gather_frac_of_hbm_peak"])
PY
