#!/usr/bin/env python
"""Kernel timeline of the last graph-replayed train steps from a rocprofv3 --kernel-trace CSV: start / end of every
kernel relative to the first one shown, with its hardware queue, so that overlap, cross-queue gaps and the gap
between two graph launches are visible.     python tools/timeline.py <..._kernel_trace.csv> [n_steps] [skip_last]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # trailing tower launches to leave out (bench.py's eager in-step probe)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"])
            for r in rows)


def short(n):
    m = re.search(r"(k_\w+|rccl\w+|Cijk\w+|at::native::\w+|__amd\w+)", n)
    return (m.group(1) if m else n)[:28]


tr = [i for i, e in enumerate(ev) if "k_mlp_train" in e[2] or "k_embed_tower_train" in e[2]]
fused = any("k_embed_tower_train" in e[2] for e in ev)      # round 4: the gather runs inside the tower launch
if len(tr) < n_steps + 3:
    sys.exit("not enough train steps in the trace")
if skip:
    tr = tr[:-skip]
lo, hi = tr[-(n_steps + 2)], tr[-3]
while not fused and lo > 0 and "k_embed_fwd" not in ev[lo][2]:
    lo -= 1
t0 = ev[lo][0]
for s, e, n, q, st in ev[lo:hi + 8]:
    print("%8.1f -> %8.1f (%5.1f us) q=%s s=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, short(n)))
starts = [ev[i][0] for i in tr]
print("step periods (us):", [round((b - a) / 1e3) for a, b in zip(starts, starts[1:])][-40:])
