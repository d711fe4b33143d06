#!/bin/bash
# round 4: tower micro-changes (FM sums with all LDS reads in flight, dnn_linear weights of the backward requested early)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_6
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_step_engine.py tests/test_gpu_mlp.py -q --tb=short 2>&1 | tail -6) > $O/pytest.log
B="--no-other-configs --no-cpu-baseline --steps 200 --warmup 20"
timeout 300 python bench.py $B > $O/bench_engine.json 2> $O/bench_engine.err
timeout 300 python bench.py $B --steps 100 --warmup 10 --diag-trace $O/trace.npy > $O/bench_diag.json 2> $O/bench_diag.err
python tools/tower_trace.py $O/trace.npy > $O/trace.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_6/bench_engine.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["value"])
t=json.load(open("gpurun_out/r4_6/trace.json"))["train"]
for k in ("fwd_stage_x","fwd_gather_round_trip_1","fwd_layer0_mfma","fwd_layer0_epilogue","fwd_layer1_mfma","fwd_projection","head","bwd_stage_top","bwd_layer1_mfma+epi","bwd_layer0_mfma+epi","wg_total"): print(k, t[k]["mean"])
PY
tail -n 3 $O/pytest.log
