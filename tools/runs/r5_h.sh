#!/bin/bash
# round 5: max pooling inside the tower launch's gather stage -- engine parity tests, headline + pooled legs
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_step_engine.py tests/test_gpu_update_general.py tests/test_gpu_deepfm.py tests/test_gpu_reference_matrix.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > $O/pytest.txt
tail -4 $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --legs deepfm_varlen --no-saturating > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_h/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'], d['roofline']['frac'], (d['roofline'].get('dominant') or {}).get('avg_us'))
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('ms_per_step'), v.get('error'), 'engine', v.get('step_engine'))
PY
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-saturating --ids zipf 2> /dev/null | grep '^{' > $O/bench_zipf.json
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-saturating --optimizer sgd 2> /dev/null | grep '^{' > $O/bench_sgd.json
python -c "
import json
for f in ('bench_zipf','bench_sgd'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'])"
