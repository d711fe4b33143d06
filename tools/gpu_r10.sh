#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_models.py tests/test_gpu_parallel.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error" $OUT/pytest_gpu.log | tail -6
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline ) 2> $OUT/bench.err | grep '^{' > $OUT/bench.json; echo "bench rc=$?"; tail -3 $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'])"
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline --force-parallel ) 2> /dev/null | grep '^{' > $OUT/bench_shard1.json
python -c "import json;d=json.load(open('$OUT/bench_shard1.json'));print('shard1', d['value'],d['ms_per_step'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o deepfm -- python $OUT/../bench.py --steps 100 --warmup 12 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1
