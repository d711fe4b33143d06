#!/bin/bash
# sharded step (direct exchange) with the pre-sorted owners' update: N-process parity, one-rank timing with / without
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_6
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_direct_exchange.py tests/test_gpu_shard_kernels.py tests/test_gpu_parallel.py -q --tb=short 2>&1 | grep -v "amdgpu.ids\|Gloo\|CudaIPC" | tail -30 > $O/pytest.txt
run() {  # tag, env...
  tag=$1; shift
  env MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 "$@" timeout 600 python bench.py --gpus 1 --force-parallel --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --repeats 3 2> $O/bench_$tag.err | grep '^{' > $O/bench_$tag.json
  python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag',d['value'],d['ms_per_step'])" >> $O/summary.txt
}
run block DCTR_SHARDED_EXCHANGE=direct
run single DCTR_SHARDED_EXCHANGE=direct DCTR_SHARDED_BLOCK=0
cat $O/summary.txt; tail -n 25 $O/pytest.txt; tail -n 3 $O/bench_block.err
