#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py -m gpu -q --tb=short -p no:cacheprovider -x ) > $OUT/pytest_r15.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" $OUT/pytest_r15.log | tail -12
for spg in 4 8; do
( timeout 300 python bench.py --steps 208 --warmup 24 --no-cpu-baseline --steps-per-graph $spg ) 2> $OUT/bench.err | grep '^{' > $OUT/bench_r15_$spg.json; echo "bench spg=$spg rc=$?"; tail -2 $OUT/bench.err | grep -v amdgpu.ids
python -c "
import json;d=json.load(open('$OUT/bench_r15_$spg.json'));print(d['value'],d['ms_per_step'],d['final_loss'])"
done
rm -rf $OUT/tl; ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $OUT/../bench.py --steps 48 --warmup 12 --no-cpu-baseline --kernel-iters 2 ) > /dev/null 2>&1
echo trace done
