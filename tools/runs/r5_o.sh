#!/bin/bash
# Adam scalar tables in the lazy update: the lazy / adam GPU tests, then the default_kwargs leg with per-kernel stats
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_o
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -x -k "lazy or adam or Adam or default or golden" > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/bench_leg.py default_kwargs --steps 100 > $O/run.log 2>&1
S=$(find $O/prof -name '*kernel_stats.csv' | head -1)
cp $S $O/default_kwargs_kernel_stats.csv
rm -rf $O/prof
head -8 $O/default_kwargs_kernel_stats.csv | cut -c1-160
grep '^{' $O/run.log | cut -c1-400
