#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3_11
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_deepfm.py tests/test_gpu_mlp.py -x -q 2>&1 | tail -5) > $O/pytest.log
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_deepfm.json 2> $O/bench_deepfm.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --ids zipf > $O/bench_zipf.json 2> $O/bench_zipf.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 96 --warmup 16 --kernel-iters 5) > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $O/deepfm_kernel_stats.csv
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 9 > $O/timeline.txt
