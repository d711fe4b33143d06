#!/bin/bash
# xDeepFM after a change to its autograd glue: the tests that cover it, then the leg with its launch table
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/xdeepfm_check; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dense_multi.py tests/test_gpu_cin.py tests/test_gpu_glue.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests_a.log
timeout 1500 python -m pytest tests -x -q -m gpu -k "xdeepfm or xDeepFM or cin or CIN" 2>&1 | tail -3 | tee $O/tests_b.log
bash tools/runs/leg.sh xdeepfm
