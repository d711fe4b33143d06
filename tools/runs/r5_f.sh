#!/bin/bash
# round 5, final-tree check: smoke + the whole GPU suite on poisoned memory, the bench line with the driver's flags
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
( timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -n 6 $O/pytest_gpu_full.log
( timeout 400 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_driver.err | grep '^{' > $O/bench_driver_flags.json
python -c "
import json
d=json.load(open('$O/bench_driver_flags.json')); print('driver flags:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], {k:(round(v['ms_per_step'],4) if 'ms_per_step' in v else v.get('error')) for k,v in d.get('other_configs',{}).items()}); print(d['cpu_baseline']['value'], d['cpu_baseline']['threads'])"
