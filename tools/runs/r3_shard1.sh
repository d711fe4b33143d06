#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_shard1
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parallel.py -x -q 2>&1 | grep -E "passed|failed|Error" > $O/pytest.txt
timeout 300 python tools/shard_host_profile.py > $O/host_profile.txt 2> $O/host_profile.err
for v in 1 0 1 0; do
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 DCTR_SHARDED_GRAPH_ALL=$v timeout 600 python bench.py --gpus 1 --force-parallel --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --repeats 3 2> $O/bench.err | grep '^{' > $O/bench_sharded_1rank_$v.json
python -c "import json;d=json.load(open('$O/bench_sharded_1rank_$v.json'));print('graph_all=$v',d['value'],d['ms_per_step'])" >> $O/summary.txt
done
