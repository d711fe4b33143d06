#!/bin/bash
# glue launches (dctr_rows_join, dctr_relu_bwd_bias): kernel tests, model parity, FiBiNET / xDeepFM step time with / without
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_7
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_full_golden.py -q --tb=short -x 2>&1 | tail -6) | tee $O/pytest1.log
(timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference_matrix.py -q --tb=short -x -k "fibinet or FiBiNET or xdeepfm or xDeepFM or DCN or PNN" 2>&1 | tail -4) | tee $O/pytest2.log
for g in 1 0; do
DCTR_GLUE_KERNELS=$g timeout 600 python - > $O/other_glue$g.json 2> $O/other_glue$g.err <<'PY'
import sys, json
sys.argv=["bench.py"]
sys.path.insert(0, ".")
import bench, torch
a = bench.parse()
a.steps_per_graph = bench.auto_steps_per_graph(a.steps)
X, y = bench.synth(a, "cuda:0", 0)
for name in ("fibinet", "xdeepfm"):
    d = bench.other_config(name, a, "cuda:0", X, y)
    print(json.dumps({k: v for k, v in d.items() if k in ("ms_per_step", "value", "error", "config")} | {"name": name}))
PY
echo "glue=$g"; cat $O/other_glue$g.json | cut -c1-200
done
