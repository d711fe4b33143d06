#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_deepfm.py tests/test_gpu_mlp.py tests/test_gpu_parallel.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_gpu.log
timeout 200 python tools/upd_trace.py > $OUT/upd_trace.json 2> $OUT/upd_trace.err; tail -2 $OUT/upd_trace.err
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline ) 2> $OUT/bench.err | grep '^{' > $OUT/bench.json; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['roofline'])"
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --optimizer sgd ) 2> $OUT/bench_sgd.err | grep '^{' > $OUT/bench_sgd.json
python -c "import json;d=json.load(open('$OUT/bench_sgd.json'));print('sgd',d['value'],d['ms_per_step'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o deepfm -- python $OUT/../bench.py --steps 100 --warmup 10 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
