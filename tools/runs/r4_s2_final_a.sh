#!/bin/bash
# end of round 4: what the driver runs (smoke, the whole GPU suite) on POISONED memory (tools/probes/poison_vram.py)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probes/poison_vram.py 2>&1 | tail -1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
( timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider ) > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -n 6 $O/pytest_gpu_full.log
