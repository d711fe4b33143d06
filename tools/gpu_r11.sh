#!/bin/bash
# round-1 session 5: CIN / bilinear kernel rewrites -- parity tests, per-kernel stats of xDeepFM and FiBiNET, step times
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( timeout 600 python -m pytest tests/test_gpu_cin.py tests/test_gpu_pairwise.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/pytest_r11.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" $OUT/pytest_r11.log | tail -12
for m in xDeepFM FiBiNET; do
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o $m -- python $OUT/../tools/prof_one_model.py $m ) > $OUT/rocprof_$m.log 2>&1; echo "rocprof $m rc=$?"
  f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %4s avg_us %9.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
( timeout 400 python tools/bench_models.py ) > $OUT/models.json 2> $OUT/models.err; echo "models rc=$?"
python -c "
import json
d=json.load(open('$OUT/models.json'))
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
