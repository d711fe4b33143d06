#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counters: python tools/pmc_kernels.py <dir> <name substring> [...]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for s in subs:
            if s in n:
                acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
for s in subs:
    print(s, {k: round(sum(v) / len(v), 1) for k, v in sorted(acc[s].items())}, "n=%d" % max([len(v) for v in acc[s].values()] or [0]))
