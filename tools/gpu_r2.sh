#!/bin/bash
# One gpurun call: GPU parity tests, bench (graph + eager), rocprof kernel stats.  Everything goes to gpurun_out/.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_models.py tests/test_gpu_update.py tests/test_gpu_parallel.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -40 $OUT/pytest_gpu.log
( timeout 600 python bench.py --steps 200 --warmup 20 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.log
cat $OUT/bench.json; tail -5 $OUT/bench.err
( timeout 300 python bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline ) > $OUT/bench_eager.json 2> $OUT/bench_eager.err
cat $OUT/bench_eager.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o deepfm -- python $OUT/../bench.py --steps 100 --warmup 10 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/summary.log
f=$(find $OUT/prof -name "*kernel_stats*" | head -1); echo $f; head -30 "$f" | cut -c1-200
