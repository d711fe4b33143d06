#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -k "fibinet or FiBiNET or bilinear or pairwise or senet or SENET" 2>&1 | tail -3
bash tools/runs/leg.sh fibinet 2>&1 | grep -E "^fibinet|senet|reduce_partials"
