#!/bin/bash
# HBM traffic of the hand-written kernels from the L2's memory-side counters.  FETCH_SIZE (3 TCC slots) and
# WRITE_SIZE (2) do not fit one pass, and a --pmc run must not be combined with other trace domains: two passes,
# each with --kernel-trace only.  Output: gpurun_out/pmc/{fetch,write}/... and gpurun_out/pmc_summary.json.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $REPO/tools/pmc_driver.py > $OUT/$c.log 2>&1
  echo "pmc $c rc=$?"
  # calibration of the UPDATE kernel's own access pattern: random 128-byte lines read-modify-written (and 64 / 128-byte
  # random reads) with known byte counts -- tools/micro/rowbench.hip
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c/rowbench -o pmc -- $REPO/tools/micro/rowbench > $OUT/${c}_rowbench.log 2>&1
  echo "pmc $c (rowbench) rc=$?"
done
cd $REPO
python tools/pmc_summary.py $OUT > gpurun_out/pmc_summary.json
cat gpurun_out/pmc_summary.json
