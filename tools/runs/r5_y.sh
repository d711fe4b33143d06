#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_y
mkdir -p $O
cd $GRAFT_REPO_ROOT
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall seconds: $SECONDS"
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_y/bench.json') if l.startswith('{')][-1])
keys=['metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data']
print({k:d[k] for k in keys})
print(d['config']['workload'])
r=d['roofline']; print({k:r[k] for k in ('bound','kernel','achieved','peak','unit','frac','traffic','frac_8d','whole_step_frac_8d')})
print('dominant', {k:r['dominant'][k] for k in ('kernel','bound','frac','avg_us')})
c=d['cpu_baseline']; print({k:c[k] for k in ('value','unit','cores','threads','kind')}, c['thread_sweep_samples_per_s'])
for k,v in d['other_configs'].items(): print(k, round(v.get('ms_per_step',-1),4), v.get('error'))
s=d['hot_path']['saturating']; print(s['gather_frac_of_hbm_peak'], s['update_path_frac_of_hbm_peak'], {k:round(v['frac_of_hbm_peak'],3) for k,v in s['embed_fwd_forward_only'].items() if isinstance(v,dict)})
PY
