#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_par
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_shard_kernels.py tests/test_gpu_parallel.py tests/test_gpu_checkpoint.py tests/test_abi.py -q --tb=short 2>&1 | tail -30) > $O/pytest.log
( timeout 300 python bench.py --steps 200 --warmup 24 --no-cpu-baseline --force-parallel ) 2> $O/shard1.err | grep '^{' > $O/bench_shard1.json
