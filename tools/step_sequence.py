#!/usr/bin/env python
"""Kernels of ONE late train step, in launch order, from a rocprofv3 --kernel-trace CSV (which glue sits where).
    python tools/step_sequence.py <..._kernel_trace.csv> [step_from_end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
fw = [i for i, e in enumerate(ev) if "k_embed_fwd" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = fw[-k - 1], fw[-k]
t0 = ev[lo][0]
for s, e, n in ev[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)
    print("%8.1f %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:150]))
