#!/bin/bash
# per-kernel durations of the two-level pre-pass at B = 262 144 / 32 768
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_3
mkdir -p $O
cd /tmp; rm -rf /tmp/prof_p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o pp -- python $GRAFT_REPO_ROOT/tools/prepass_bench.py 32768 262144 > $O/prof.log 2>&1
f=$(find /tmp/prof_p -name "*kernel_stats.csv" | head -1); cp $f $O/prepass_kernel_stats.csv
t=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
python - $t <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "prepass" in n or "k_bucket" in n or "k_embed_segments" in n:
        agg[(n.split("(")[0][-40:], r.get("Grid_Size") or r.get("Grid_Size_X") or "")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items()):
    v=sorted(v); print(k, "n=%d median %.1f us min %.1f" % (len(v), v[len(v)//2], v[0]))
PY
cd $GRAFT_REPO_ROOT
timeout 300 python tools/prepass_bench.py > $O/prepass_uniform.jsonl 2>/dev/null
timeout 300 python tools/prepass_bench.py --zipf > $O/prepass_zipf.jsonl 2>/dev/null
cat $O/prepass_uniform.jsonl $O/prepass_zipf.jsonl | cut -c1-300
(timeout 600 python -m pytest tests/test_gpu_update.py -q --tb=short -x 2>&1 | tail -3)
