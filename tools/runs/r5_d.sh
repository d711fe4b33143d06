#!/bin/bash
# round 5, fourth GPU pass: smoke + the whole suite on POISONED memory, PMC traffic of the renamed update kernels, rocprofv3
# kernel stats + timeline of the bench command, fit() host profile, the bench line with the driver's flags
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -n 8 $O/pytest_gpu_full.log
PMC_OPTS=adagrad PMC_BATCHES=4096 bash tools/pmc_traffic.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-300
cp gpurun_out/pmc_summary.json $O/pmc_summary.json
rm -rf /tmp/prof5
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 25 --no-cpu-baseline --no-other-configs --no-saturating ) > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find /tmp/prof5 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv; head -8 $f | cut -c1-160
t=$(find /tmp/prof5 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 90 > $O/timeline.txt 2>&1; tail -30 $O/timeline.txt | cut -c1-160
( timeout 300 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_driver.err | grep '^{' > $O/bench_driver_flags.json
python -c "
import json
d=json.load(open('$O/bench_driver_flags.json')); print('driver flags:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], {k:(round(v['ms_per_step'],4) if 'ms_per_step' in v else v.get('error')) for k,v in d.get('other_configs',{}).items()})"
( timeout 300 python tools/fit_profile.py ) > $O/fit_profile.txt 2>&1; head -12 $O/fit_profile.txt; grep "one epoch\|40 groups" $O/fit_profile.txt
