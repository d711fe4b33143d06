#!/bin/bash
# fork/join of the tower weight gradients: parity + bench A/B
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_models.py tests/test_gpu_parallel.py -m gpu -q --tb=short -p no:cacheprovider -x ) > $OUT/pytest_r13.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" $OUT/pytest_r13.log | tail -12
for ov in 1 0; do
( DCTR_OVERLAP_WGRAD=$ov timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline ) 2> $OUT/bench.err | grep '^{' > $OUT/bench_r13_$ov.json; echo "bench overlap=$ov rc=$?"; tail -2 $OUT/bench.err
python -c "
import json;d=json.load(open('$OUT/bench_r13_$ov.json'));print(d['value'],d['ms_per_step'],d['final_loss'])"
done
( DCTR_OVERLAP_WGRAD=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph ) 2> /dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('eager overlap', d['value'],d['ms_per_step'])"
