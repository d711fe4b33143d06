#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_parallel.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -4
for ov in 1 0; do
  DCTR_OVERLAP_WGRAD=$ov python bench.py --steps 200 --warmup 24 --no-cpu-baseline --force-parallel 2>/dev/null | grep '^{' > $OUT/bench_shard_ov$ov.json
  python -c "
import json;d=json.load(open('$OUT/bench_shard_ov$ov.json'));print('overlap', $ov, d['value'], d['ms_per_step'], d['final_loss'])"
done
