#!/bin/bash
# round 5, second GPU pass: general units + pooled fields inside the engine; varlen leg timed and profiled
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_update_general.py tests/test_gpu_step_engine.py tests/test_gpu_update.py tests/test_gpu_deepfm.py tests/test_gpu_full_golden.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -80 > $O/pytest.txt
tail -5 $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --legs deepfm_varlen,default_kwargs --no-saturating > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_b/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'])
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('ms_per_step'), v.get('error'), 'engine', v.get('step_engine'), 'unit_path', v.get('unit_path'), v.get('ms_per_step_vs_headline'))
PY
cd /tmp; rm -rf /tmp/prof_v
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o vl -- python $GRAFT_REPO_ROOT/tools/bench_leg.py deepfm_varlen --steps 100 --repeats 1 --warmup-seconds 0 > $O/leg_prof.json 2> $O/leg_prof.err
f=$(find /tmp/prof_v -name "*kernel_stats.csv" | head -1); cp $f $O/varlen_kernel_stats.csv; head -12 $f
