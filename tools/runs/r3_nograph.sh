#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_nograph
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 64 --warmup 8 --repeats 1 --no-graph --diag-trace $O/trace_eager.npy > $O/bench_eager.json 2> $O/bench_eager.err
python tools/tower_trace.py $O/trace_eager.npy > $O/trace_eager.json
DCTR_SEGMENTS=0 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 1 --diag-trace $O/trace_noseg.npy > $O/bench_noseg.json 2> $O/bench_noseg.err
python tools/tower_trace.py $O/trace_noseg.npy > $O/trace_noseg.json
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 200 --warmup 20 --repeats 1 --diag-trace $O/trace_base.npy > $O/bench_base.json 2> $O/bench_base.err
python tools/tower_trace.py $O/trace_base.npy > $O/trace_base.json
