#!/bin/bash
# end-of-round measurement set: smoke, full GPU suite, bench (+cpu baseline, sgd, eager, sharded 1-rank), rocprofv3 kernel
# stats + timeline of the bench command, the other models' step times, the default-kwargs (lazy vs dense) comparison.
# Heavier, optional: PMC traffic (tools/pmc_traffic.sh), microbench, phase traces -- run separately when kernels change.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_gpu_full.log | tail -2
( timeout 600 python bench.py ) 2> $OUT/bench.err | grep '^{' > $OUT/bench.json; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'],d['cpu_baseline']['cores'])"
( timeout 300 python bench.py --steps 20 --warmup 5 ) 2> /dev/null | grep '^{' > $OUT/bench_driver_flags.json     # what the driver runs
( timeout 300 python bench.py --no-cpu-baseline --no-other-configs --ids zipf ) 2> /dev/null | grep '^{' > $OUT/bench_zipf.json
python -c "
import json
for f in ('bench_driver_flags','bench_zipf'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d.get('other_configs',{}).items()})"
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline --optimizer sgd ) 2> /dev/null | grep '^{' > $OUT/bench_sgd.json
( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph ) 2> /dev/null | grep '^{' > $OUT/bench_eager.json
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline --force-parallel ) 2> /dev/null | grep '^{' > $OUT/bench_shard1.json
python -c "
import json
for f in ('bench_sgd','bench_eager','bench_shard1'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
rm -rf $OUT/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o deepfm -- python $OUT/../bench.py --steps 96 --warmup 16 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-140
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 > $OUT/timeline.txt; tail -1 $OUT/timeline.txt
cp $f $OUT/deepfm_kernel_stats.csv; rm -rf $OUT/prof       # (gpurun copies back at most 64 MiB: the raw trace is not kept)
( timeout 600 python tools/bench_models.py ) > $OUT/models.json 2> $OUT/models.err; echo "models rc=$?"
python -c "
import json
for k,v in json.load(open('$OUT/models.json')).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
( timeout 600 python tools/bench_defaults.py ) > $OUT/defaults.json 2> $OUT/defaults.err; echo "defaults rc=$?"
python -c "
import json
for k,v in json.load(open('$OUT/defaults.json')).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
