#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_cin
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin.py tests/test_gpu_dense_multi.py tests/test_gpu_models.py -x -q -k "cin or xdeepfm or pool" 2>&1 | tail -3 > $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_full_golden.py -x -q -k xdeepfm 2>&1 | tail -3 >> $O/pytest.txt
cd /tmp
for m in xDeepFM; do
rm -rf /tmp/p_$m
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$m -o m -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py $m > $O/$m.log 2>&1
t=$(find /tmp/p_$m -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/step_profile.py $t 1 4 > $O/budget_$m.txt 2>&1
done
