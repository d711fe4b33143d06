#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_gpu_full.log | tail -3
( timeout 600 python tools/bench_defaults.py ) > $OUT/defaults.json 2> $OUT/defaults.err; echo "defaults rc=$?"; tail -2 $OUT/defaults.err | grep -v amdgpu
cat $OUT/defaults.json
