#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench, microbench, rocprof kernel trace.  Everything goes to gpurun_out/.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.log
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.log
tail -5 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 200 --warmup 20 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.log
cat $OUT/bench.json
( timeout 600 python bench.py --steps 200 --warmup 20 --optimizer sgd --no-cpu-baseline ) > $OUT/bench_sgd.json 2> $OUT/bench_sgd.err
( timeout 600 python bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline ) > $OUT/bench_eager.json 2> $OUT/bench_eager.err
( timeout 900 python tools/microbench.py ) > $OUT/microbench.json 2> $OUT/microbench.err; echo "microbench rc=$?" | tee -a $OUT/summary.log
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o deepfm -- python $OLDPWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/summary.log
find $OUT/prof -name "*stats*" | head
