#!/bin/bash
# the CIN stack (one autograd node, projection folded in): its tests, the atomic-fallback test, xDeepFM's bench leg
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin.py tests/test_gpu_update_general.py tests/test_gpu_models.py tests/test_gpu_full_golden.py -q -m gpu -x -k "cin or xdeepfm or xDeepFM or past_the_envelope or golden" > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
timeout 600 python tools/bench_leg.py xdeepfm > $O/xdeepfm.json 2> $O/xdeepfm.err
tail -3 $O/xdeepfm.json; tail -3 $O/xdeepfm.err
