#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_gaps
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_deepfm.py -x -q 2>&1 | tail -15 > $O/pytest.txt
