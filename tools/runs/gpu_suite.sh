#!/bin/bash
# the whole GPU suite + smoke (what the driver runs at round end)
export TMPDIR=/tmp
O=gpurun_out/suite
mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"
( timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu_full.log
