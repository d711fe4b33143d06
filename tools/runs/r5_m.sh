#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_m
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/probes/aten_in_step.py xDeepFM > $O/aten_xdeepfm.txt 2>&1
cat $O/aten_xdeepfm.txt | tail -60
