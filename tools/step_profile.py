#!/usr/bin/env python
"""Per-step kernel budget of a model from a rocprofv3 --kernel-trace CSV: the window between two gathers (k_embed_fwd)
late in the run, aggregated by kernel -- launches per step, µs per step, share.  (The --stats summary mixes the train
steps with model construction.)      python tools/step_profile.py <..._kernel_trace.csv> [first_step] [n_steps] [--order]
--order: also the launches of the window's first step in start order (name, start, duration, queue)."""
import csv
import re
import sys
from collections import defaultdict

order = "--order" in sys.argv
if order:
    sys.argv.remove("--order")
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
fw = [i for i, e in enumerate(ev) if "k_embed_fwd" in e[2]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(fw) - 12)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lo, hi = fw[first], fw[first + n]
agg = defaultdict(lambda: [0, 0])


def short(name):
    m = re.search(r"(k_\w+|Cijk_\w+?_MT\w+?_|multi_tensor_apply_kernel|CatArrayBatchedCopy|reduce_kernel|"
                  r"vectorized_elementwise_kernel<\d+, at::native::\w+|elementwise_kernel\w*|__amd_rocclr_\w+|"
                  r"at::native::\w+)", name)
    return (m.group(1) if m else name)[:70]


for s, e, name, _q in ev[lo:hi]:
    a = agg[short(name)]
    a[0] += 1
    a[1] += e - s
span = (ev[hi][0] - ev[lo][0]) / n / 1e3
busy = sum(v[1] for v in agg.values()) / n / 1e3
print("step period %.1f us, kernel time %.1f us/step, %d launches/step" % (span, busy, sum(v[0] for v in agg.values()) / n))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-70s %6.1f launches  %8.1f us  %5.1f%%" % (k, c / n, t / n / 1e3, 100.0 * t / n / 1e3 / busy))
if order:
    t0 = ev[lo][0]
    for s, e, name, q in ev[lo:fw[first + 1]]:
        print("  %8.1f  %7.1f us  q=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:150]))
