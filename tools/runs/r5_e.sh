#!/bin/bash
# round 5, fifth GPU pass: the whole suite on poisoned memory again (test fixes, permutation threads, replicated fallback),
# the default bench line, the graph-replayed timeline
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -n 8 $O/pytest_gpu_full.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5_e/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'], 'traffic', d['roofline']['traffic'])
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('ms_per_step'), v.get('error'), v.get('step_engine'))
fa=d['other_configs']['fit_api']; print({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if a in('value','ms_per_step')}) for k,v in fa.items()})
PY
rm -rf /tmp/prof5
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 25 --no-cpu-baseline --no-other-configs --no-saturating ) > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
t=$(find /tmp/prof5 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 6 90 > $O/timeline.txt 2>&1; tail -40 $O/timeline.txt | cut -c1-160
