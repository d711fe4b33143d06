#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_glue.py tests/test_gpu_dense_multi.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python -m pytest tests -x -q -m gpu -k "fibinet or FiBiNET or xdeepfm or xDeepFM or dcn or DCN" 2>&1 | tail -3
for leg in fibinet xdeepfm; do
bash tools/runs/leg.sh $leg 2>&1 | grep -E "^$leg|colsum|rows_dot"
done
