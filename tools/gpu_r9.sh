#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_deepfm.py tests/test_gpu_fullsize.py tests/test_gpu_parallel.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_gpu.log
timeout 200 python tools/upd_trace.py > $OUT/upd_trace.json 2> $OUT/upd_trace.err; tail -2 $OUT/upd_trace.err
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline ) 2> $OUT/bench.err | grep '^{' > $OUT/bench.json; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['hot_path']['kernels'])"
( timeout 600 python tools/microbench.py ) > $OUT/microbench.json 2> $OUT/microbench.err; echo "microbench rc=$?"
python -c "
import json
m=json.load(open('$OUT/microbench.json'))
for k in m:
    if k.startswith('embed_kernels'): print(k, {n:(round(v['avg_us'],1), round(v['gbs'])) for n,v in m[k].items()})
"
