#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_2
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -5) > $O/pytest_mlp.log
timeout 300 python tools/tower_bench.py --diag --slices 5,6,7,8 > $O/tower_deepfm.json 2> $O/tower_deepfm.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_deepfm.json 2> $O/bench_deepfm.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --diag-trace $O/trace.npy > $O/bench_diag.json 2> $O/bench_diag.err
python tools/tower_trace.py $O/trace.npy > $O/trace.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 > $O/timeline.txt
