#!/bin/bash
# final tree: smoke + the whole GPU suite on poisoned memory + the bench line with the driver's flags
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_final
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/pytest_gpu_full.log | tail -1
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall seconds: $SECONDS"
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_final/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['other_configs'].items(): print(k, round(v.get('ms_per_step',-1),4), v.get('error'), (v.get('steady_state') or {}).get('ms_per_step'))
PY
