#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/fit_sharded
DCTR_FIT_FORCE_TRAINER=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29411 timeout 600 python tools/probes/fit_sharded_overhead.py > gpurun_out/fit_sharded/out.txt 2>&1
head -90 gpurun_out/fit_sharded/out.txt
