#!/bin/bash
# round 4: launch order of one eager FiBiNET / xDeepFM step (which glue kernels sit between the layer kernels)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_order2
mkdir -p $O
for m in FiBiNET; do
cd /tmp; rm -rf /tmp/prof_o
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -o o -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py $m > $O/prof_$m.log 2>&1
t=$(find /tmp/prof_o -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/step_profile.py $t 1 4 --order > $O/${m}_order.txt 2>&1
done
grep -v "^  k_\|Cijk" $O/FiBiNET_order.txt | tail -60 | cut -c1-260
