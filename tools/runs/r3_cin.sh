#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_cin
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin.py tests/test_gpu_models.py -x -q -k "cin or xdeepfm" 2>&1 | grep -E "passed|failed|Error" > $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_full_golden.py -x -q -k xdeepfm 2>&1 | grep -E "passed|failed|Error" >> $O/pytest.txt
cd /tmp
for ct in 1 2; do
rm -rf /tmp/p_x
DCTR_CIN_BWD_CT=$ct timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_x -o m -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py xDeepFM > $O/x$ct.log 2>&1
t=$(find /tmp/p_x -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/step_profile.py $t 1 4 > $O/budget_ct$ct.txt 2>&1
done
