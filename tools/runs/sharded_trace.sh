#!/bin/bash
# the table-sharded step at one rank (direct exchange, S steps per hipGraph): driver-style timing + a kernel timeline
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sharded
mkdir -p $O
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
timeout 300 python bench.py --gpus 1 --force-parallel --exchange direct --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating --kernel-iters 2 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1]); print("sharded_1rank", round(d["ms_per_step"],5), d["timing"]["ms_per_step_all"], d["final_loss"])
except Exception as e: print("failed", e); print(open("$O/bench.err").read()[-1500:])
PY
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --gpus 1 --force-parallel --exchange direct --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating --kernel-iters 2 --warmup-seconds 0.2 --repeats 1 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/timeline.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
def short(n):
    m = re.search(r"(k_\w+|rccl\w+|Cijk\w+|at::native::\w+|__amd\w+)", n); return (m.group(1) if m else n)[:30]
tw = [i for i, e in enumerate(ev) if "k_embed_tower_train" in e[2] or "k_mlp_train" in e[2]]
lo, hi = tw[-8], tw[-4]
t0 = ev[lo][0]
for s, e, n, q in ev[lo - 6:hi + 2]:
    print("%8.1f -> %8.1f (%5.1f us) q=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
st = [ev[i][0] for i in tw]
big = [k for k in range(1, len(st)) if st[k] - st[k - 1] > 250e3]
if big:
    k = big[-1]
    print("---- a block boundary (between two hipGraph launches):")
    for s, e, n, q in ev[tw[k - 1]:tw[k] + 3]:
        print("%8.1f -> %8.1f (%5.1f us) q=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
print("step periods (us):", [round((b - a) / 1e3) for a, b in zip(st, st[1:])][-45:])
PY
cat $O/timeline.txt | head -90
rm -rf $O/trace
for st in 100 200; do
timeout 300 python bench.py --gpus 1 --force-parallel --exchange direct --steps $st --warmup 5 --no-other-configs --no-cpu-baseline --no-saturating --kernel-iters 2 > $O/bench_$st.json 2> $O/bench_$st.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$st.json") if l.startswith("{")][-1]); print("sharded_1rank steps=$st S=", d["config"]["steps_per_graph"], round(d["ms_per_step"],5), d["timing"]["ms_per_step_all"])
except Exception as e: print("failed", e)
PY
done
