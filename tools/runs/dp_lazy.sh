#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dp_lazy
timeout 900 python -m pytest tests/test_gpu_lazy.py -x -q -m gpu -k "data_parallel" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 900 python tools/probes/dp_lazy_step.py 2>&1 | grep -E "DCTR_DP_LAZY|Error|error" | tee gpurun_out/dp_lazy/out.txt
