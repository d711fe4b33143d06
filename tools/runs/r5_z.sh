#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_z
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -n 6 $O/pytest_gpu_full.log
