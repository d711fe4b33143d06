#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_full
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_full_golden.py tests/test_gpu_fullsize.py -q --tb=short 2>&1 | tail -40) > $O/pytest.log
