#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_t
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/probes/lazy_gaps.py 1600 > $O/gaps.txt 2>&1
tail -12 $O/gaps.txt
