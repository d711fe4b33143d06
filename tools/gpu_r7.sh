#!/bin/bash
# smoke, the complete GPU suite, PMC traffic passes, microbench
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest_gpu_full.log
timeout 900 bash tools/pmc_traffic.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -3 $OUT/pmc.log | cut -c1-300
( timeout 600 python tools/microbench.py ) > $OUT/microbench.json 2> $OUT/microbench.err; echo "microbench rc=$?"
