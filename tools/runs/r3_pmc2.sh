#!/bin/bash
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_pmc2
mkdir -p $O
timeout 200 tools/micro/rowbench > $O/rowbench.json 2> $O/rowbench.err
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_summary.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc
