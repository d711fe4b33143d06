#!/bin/bash
# update kernel load order, SENET partial reduce: parity + bench
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_deepfm.py tests/test_gpu_pairwise.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -x ) > $OUT/pytest_r12.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" $OUT/pytest_r12.log | tail -12
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline ) 2> $OUT/bench.err | grep '^{' > $OUT/bench_r12.json; echo "bench rc=$?"; tail -3 $OUT/bench.err
python -c "
import json;d=json.load(open('$OUT/bench_r12.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac']); print({k:(round(v['avg_us'],1),round(v['gbs'])) for k,v in d['hot_path']['kernels'].items()})"
( timeout 200 python tools/upd_trace.py ) > $OUT/upd_trace_r12.json 2> /dev/null; python -c "
import json;d=json.load(open('$OUT/upd_trace_r12.json'));print(json.dumps(d)[:1500])"
