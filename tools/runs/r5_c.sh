#!/bin/bash
# round 5, third GPU pass: whole suite (oracle-based layer tests, fp64 bars, forward-only layout, engine + pooled), default bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 > $O/pytest.txt
tail -8 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_c/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'])
print('roofline',{k:v for k,v in d['roofline'].items() if k in('frac','frac_8d','avg_us','whole_step_frac_8d')})
print('dominant',{k:v for k,v in (d['roofline'].get('dominant') or {}).items() if k in ('frac','avg_us')})
sat=d['hot_path']['saturating']
print('sat', {k:sat[k] for k in ('gather_frac_of_hbm_peak','update_path_frac_of_hbm_peak')}, sat.get('embed_fwd_forward_only'))
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('ms_per_step'), v.get('error'), v.get('step_engine'), v.get('unit_path'))
print('cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k in ('value','threads','thread_sweep_samples_per_s')})
PY
