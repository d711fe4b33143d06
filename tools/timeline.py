#!/usr/bin/env python
"""Kernel timeline of the last train steps from a rocprofv3 --kernel-trace CSV: start / end of every kernel relative
to the step's first kernel (k_embed_fwd), per stream / queue, so that overlap and gaps are visible.
    python tools/timeline.py <..._kernel_trace.csv> [n_steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
ev.sort()
short = lambda n: n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:28]
idx = [i for i, e in enumerate(ev) if "k_embed_fwd" in e[2]]
if len(idx) < n_steps + 2:
    sys.exit("not enough steps in the trace")
first = idx[-(n_steps + 1)]
last = idx[-1]
t0 = ev[first][0]
prev_fwd = None
for s, e, n, q, st in ev[first:last]:
    if "k_embed_fwd" in n:
        if prev_fwd is not None:
            print("---- step period %.1f us" % ((s - prev_fwd) / 1e3))
        prev_fwd = s
    print("%8.1f -> %8.1f  (%6.1f us)  q=%s s=%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, short(n)))
