#!/bin/bash
# one bench leg with and without a kernel trace: gpurun -- bash tools/runs/leg.sh <leg> [env VAR=...]
export TMPDIR=/tmp
LEG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/leg_$LEG
mkdir -p $O
cd $GRAFT_REPO_ROOT
env "$@" timeout 600 python tools/bench_leg.py $LEG --steps 20 --warmup 5 > $O/leg.json 2> $O/leg.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/leg.json") if l.startswith("{")][-1])
    print("$LEG", {k: d.get(k) for k in ("ms_per_step","value","error")}, (d.get("steady_state") or {}).get("ms_per_step"))
except Exception as e: print("failed", e); print(open("$O/leg.err").read()[-800:])
PY
rm -rf $O/trace
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python tools/bench_leg.py $LEG --steps 20 --warmup 5 --warmup-seconds 0.3 --repeats 2 > $O/trace.log 2>&1
s=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python - "$s" > $O/kernels.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|rccl\w+|Cijk\w{0,40}|at::native::\w+|__amd\w+)", n); return (m.group(1) if m else n)[:60]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-62s calls %6s avg %8.1f us  %5.1f %%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -30 $O/kernels.txt
# the same launches split by grid size (k_lazy<., 3> runs twice per step: wide pass, deep pass)
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$t" > $O/kernels_by_grid.txt <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    if not m: continue
    k = (m.group(1)[:50], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
    a = acc[k]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-52s grid %10s wg %5s calls %6d avg %8.1f us" % (k[0], k[1], k[2], n, t / n))
PY
head -12 $O/kernels_by_grid.txt
rm -rf $O/trace
