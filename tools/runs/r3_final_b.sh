#!/bin/bash
# final evidence, part B: smoke + the whole GPU suite, the bench lines, kernel stats + timeline, other models' budgets
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final
mkdir -p $O
bash tools/runs/gpu_suite.sh > $O/suite.log 2>&1
cp gpurun_out/suite/smoke.log $O/smoke.log; tail -8 gpurun_out/suite/pytest_gpu_full.log > $O/pytest_tail.txt
( timeout 900 python bench.py ) 2> $O/bench.err | grep '^{' > $O/bench_default_flags.json
( timeout 600 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_drv.err | grep '^{' > $O/bench_driver_flags.json
(cd /tmp && rm -rf /tmp/prof1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5 --repeats 1) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 6 > $O/timeline.txt; python tools/step_profile.py $t 1 4 > $O/budget_DeepFM.txt 2>&1
bash tools/runs/r3_models.sh > $O/models.log 2>&1
cp gpurun_out/r3_models/budget_xDeepFM.txt gpurun_out/r3_models/budget_FiBiNET.txt $O/
timeout 300 python tools/zipf_update_probe.py uniform > $O/update_probe.json 2> $O/update_probe.err
timeout 300 python tools/step_hops.py > $O/hops.json 2> $O/hops.err
