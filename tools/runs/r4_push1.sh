#!/bin/bash
# round 4: push-style direct exchange -- parity (N processes on one GPU), one-rank timing push vs copy, timeline
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r4_push3
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_direct_exchange.py tests/test_gpu_shard_kernels.py tests/test_gpu_parallel.py -q --tb=short 2>&1 | grep -v "amdgpu.ids\|Gloo\|CudaIPC" | tail -30 > $O/pytest.txt
DCTR_SHARDED_PUSH=0 timeout 900 python -m pytest tests/test_gpu_direct_exchange.py -q --tb=short 2>&1 | grep -v "amdgpu.ids\|Gloo\|CudaIPC" | tail -5 > $O/pytest_copy.txt
run() {  # tag, env...
  tag=$1; shift
  env MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 "$@" timeout 600 python bench.py --gpus 1 --force-parallel --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --repeats 3 2> $O/bench_$tag.err | grep '^{' > $O/bench_$tag.json
  python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag',d['value'],d['ms_per_step'])" >> $O/summary.txt
}
run push DCTR_SHARDED_EXCHANGE=direct
run copy DCTR_SHARDED_EXCHANGE=direct DCTR_SHARDED_PUSH=0
cd /tmp; rm -rf /tmp/prof_s
MASTER_ADDR=127.0.0.1 MASTER_PORT=29556 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 DCTR_SHARDED_EXCHANGE=direct timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o sh -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-parallel --steps 64 --warmup 16 --no-cpu-baseline --no-other-configs --repeats 1 --warmup-seconds 0 --kernel-iters 2 > $O/bench_prof.json 2> $O/bench_prof.err
t=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/timeline.py $t 3 > $O/timeline.txt 2>&1
cat $O/summary.txt; tail -n 12 $O/pytest.txt; tail -3 $O/pytest_copy.txt; tail -n 3 $O/bench_push.err
