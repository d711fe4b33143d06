#!/bin/bash
# shard kernels' tests + the sharded step at one rank with a timeline
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_shard_kernels.py tests/test_gpu_direct_exchange.py -x -q -m gpu 2>&1 | tail -3
bash tools/runs/sharded_trace.sh 2>&1 | head -40
bash tools/runs/sharded_trace.sh 2>&1 | grep "sharded_1rank"
