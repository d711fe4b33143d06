#!/bin/bash
# xDeepFM step: per-step kernel budget and launch order after the CIN stack
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_ab
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py FiBiNET 30 > $O/run.log 2>&1
CSV=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/step_profile.py $CSV --order > $O/fibinet_step_launch_order.txt 2>&1
rm -rf $O/prof
head -80 $O/fibinet_step_launch_order.txt
