#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parallel.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_gpu.log
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --force-parallel ) 2> $OUT/bench_shard1.err | grep '^{' > $OUT/bench_shard1.json; echo "bench shard rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_shard1.json'));print(d['value'],d['ms_per_step'],d['config']['hip_graph'])"; tail -3 $OUT/bench_shard1.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_shard -o deepfm -- python $OUT/../bench.py --steps 60 --warmup 10 --no-cpu-baseline --force-parallel ) > $OUT/rocprof_shard.log 2>&1; echo "rocprof rc=$?"
