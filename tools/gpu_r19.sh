#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_parallel.py -m gpu -q --tb=short -p no:cacheprovider -x ) > $OUT/pytest_r19.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert " $OUT/pytest_r19.log | tail -8
( timeout 600 python tools/bench_defaults.py ) > $OUT/defaults.json 2> $OUT/defaults.err; echo "defaults rc=$?"
python -c "
import json
for k,v in json.load(open('$OUT/defaults.json')).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
