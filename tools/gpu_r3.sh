#!/bin/bash
# tests of the touched kernels, bench, rocprof trace of one step
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_update.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest_gpu.log
( timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o deepfm -- python $OUT/../bench.py --steps 100 --warmup 10 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
