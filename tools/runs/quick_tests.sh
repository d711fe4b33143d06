#!/bin/bash
# gpurun -- bash tools/runs/quick_tests.sh <pytest args...>      (GPU tests named on the command line, output kept)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/quick_tests
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest "$@" -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
echo "rc=$?"
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR" $O/pytest.txt | head -30
