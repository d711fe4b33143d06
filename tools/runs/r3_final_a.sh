#!/bin/bash
# final evidence, part A: PMC traffic (+ rowbench calibration), SQ / MFMA counters of the tower kernels, matrix-pipe microbenchmark
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final
mkdir -p $O
timeout 200 tools/micro/rowbench > $O/rowbench.json 2> $O/rowbench.err
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_summary.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc
bash tools/runs/r3_pmc1.sh > $O/pmc1.log 2>&1
cat gpurun_out/r3_pmc1/p1.txt gpurun_out/r3_pmc1/p2.txt gpurun_out/r3_pmc1/p3.txt > $O/tower_sq_counters.txt
timeout 200 tools/micro/mfmabench > $O/mfmabench.jsonl 2> $O/mfmabench.err
