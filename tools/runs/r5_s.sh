#!/bin/bash
# default_kwargs leg under --kernel-trace: per-kernel durations in the LAST 200 steps (the steady-state phase)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_x
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/bench_leg.py default_kwargs --steps 100 > $O/run.log 2>&1
CSV=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python - $CSV <<'PY' > $O/steady_kernels.txt
import csv,sys,re
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 200 steps = last 200 launches of the apply kernel
idx=[i for i,r in enumerate(rows) if "k_lazy<4, 1>" in r["Kernel_Name"]]
lo=idx[-200]
agg=defaultdict(lambda:[0,0])
for r in rows[lo:]:
    m=re.search(r"(k_\w+(<[^>]*>)?|Cijk_\w{0,30}|rocblas_\w+|vectorized_elementwise_kernel<\d+, at::native::\w+|elementwise_kernel\w*|__amd_rocclr_\w+|at::native::\w+)", r["Kernel_Name"])
    k=(m.group(1) if m else r["Kernel_Name"])[:60]
    agg[k][0]+=1; agg[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
span=(int(rows[-1]["End_Timestamp"])-int(rows[lo]["Start_Timestamp"]))/200/1e3
print("per step: span %.1f us"%span)
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print("  %-60s %5.2f launches %8.1f us"%(k,c/200,t/200/1e3))
PY
rm -rf $O/prof
cat $O/steady_kernels.txt
