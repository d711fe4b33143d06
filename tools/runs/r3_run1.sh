#!/bin/bash
# round 3, GPU call 1: the reworked tower kernels -- parity first, then timings
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_1
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -15) > $O/pytest_mlp.log
(timeout 600 python -m pytest tests/test_gpu_deepfm.py -x -q 2>&1 | tail -8) > $O/pytest_deepfm.log
timeout 300 python tools/tower_bench.py --trace --slices 4,5,6,7,8,9,10,12,14 > $O/tower_deepfm.json 2> $O/tower_deepfm.err
timeout 300 python tools/tower_bench.py --shape xdeepfm --diag > $O/tower_xdeepfm.json 2> $O/tower_xdeepfm.err
timeout 120 tools/micro/mfmabench > $O/mfmabench.jsonl 2>&1
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_deepfm.json 2> $O/bench_deepfm.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 > $O/timeline.txt
ls -la $O
