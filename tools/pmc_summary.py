#!/usr/bin/env python
"""Reduce the two rocprofv3 --pmc passes of tools/pmc_traffic.sh to per-kernel HBM bytes per launch.

Counter unit: KiB (hbm_bytes = counter * 1024, cdna_hip_programming.md section 7).  gfx950 correction
(MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes, i.e. reports half the bytes of a wide
coalesced read; other access patterns are "uncalibrated", so the correction factors are MEASURED here on
calibration launches with known byte counts (tools/pmc_driver.py) and reported next to the raw numbers:
    fetch_factor_stream = known / raw on the streaming add and the reduction,
    fetch_factor_gather = known / raw on the random 64-byte-row gather (the embedding kernels' pattern),
    write_factor        = known / raw on the streaming add.
Embedding kernels are corrected with the gather factor for reads and the stream factor for writes."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("deepctr-torch_amd/csrc/update.hip", "deepctr-torch_amd/csrc/update_kernels.hpp",
                  "deepctr-torch_amd/csrc/update_launch.inc", "deepctr-torch_amd/csrc/embed.hip",
                  "deepctr-torch_amd/csrc/common.hpp", "deepctr-torch_amd/csrc/lazy_opt.hpp")


def code_hash():
    """sha256 over the sources of the measured kernels: bench.py recomputes it and refuses a summary taken on other code"""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


root = sys.argv[1]
MiB = 1024 * 1024


def load(counter):
    per = defaultdict(list)   # kernel name -> [(grid, value)]
    for path in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                per[row["Kernel_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
    return per


def short(name):
    import re
    m = re.search(r"k_embed_apply_sorted<\d+, \d+, (\d)(?:, (?:false|true))?>", name)
    if m:      # the update proper (after the segment pre-pass); last template argument = the optimizer
        return {"0": "embed_update_sgd", "1": "embed_update_adagrad"}.get(m.group(1), "embed_update_accum")
    m = re.search(r"k_rows<(\d+), (\d+), (true|false), (\d+)>", name)
    if m:      # tools/micro/rowbench.hip: LPR lanes of 16 B per row, RMW or read-only, NARR arrays
        return "calib_rows_%dB_%s%s" % (16 * int(m.group(1)), "rmw" if m.group(3) == "true" else "read",
                                       "_x2arrays" if m.group(4) == "2" else "")
    if "k_embed_segments" in name:
        return "embed_segments"
    m = re.search(r"k_embed_update<\d+, \d+, (\d)(?:, (?:false|true))?>", name)
    if m:      # the general kernel (no pre-pass); last template argument = the optimizer (0 SGD, 1 Adagrad, 2 accumulate)
        return {"0": "embed_update_general_sgd", "1": "embed_update_general_adagrad"}.get(m.group(1), "embed_update_general_accum")
    for key, tag in (("k_embed_fwd", "embed_fwd"), ("k_embed_update", "embed_update"),
                     ("CUDAFunctorOnSelf_add", "calib_stream"), ("sum_functor", "calib_reduce"),
                     ("vectorized_gather_kernel", "calib_gather")):
        if key in name:
            return tag
    return None


raw = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    groups = defaultdict(list)
    for name, vals in load(counter).items():
        tag = short(name)
        if tag is None:
            continue
        for grid, v in vals:
            groups[(tag, grid)].append(v)
    raw[counter] = {"%s@grid%d" % k: {"launches": len(v), "mean_KiB": sum(v) / len(v), "max_KiB": max(v)}
                    for k, v in sorted(groups.items())}


def biggest(counter, tag):
    best = None
    for k, v in raw[counter].items():
        if k.startswith(tag + "@") and (best is None or v["mean_KiB"] > best["mean_KiB"]):
            best = v
    return best


out = OrderedDict()
out["code_hash"] = code_hash()
out["code_hash_of"] = list(KERNEL_SOURCES)
out["unit_note"] = "raw counters are KiB per launch; *_bytes are corrected bytes per launch"
known = {"calib_stream": (256 * MiB, 256 * MiB), "calib_reduce": (256 * MiB, 0),
         "calib_gather": (64 * MiB + 8 * MiB, 64 * MiB)}
fac = {}
for tag, (rd, wr) in known.items():
    f, w = biggest("FETCH_SIZE", tag), biggest("WRITE_SIZE", tag)
    if f is None and w is None:
        continue
    fac[tag] = {"fetch_raw_bytes": f["mean_KiB"] * 1024 if f else None, "fetch_known_bytes": rd,
                "fetch_factor": (rd / (f["mean_KiB"] * 1024)) if f and f["mean_KiB"] else None,
                "write_raw_bytes": w["mean_KiB"] * 1024 if w else None, "write_known_bytes": wr,
                "write_factor": (wr / (w["mean_KiB"] * 1024)) if w and w["mean_KiB"] and wr else None}
# the update kernel's own pattern (tools/micro/rowbench.hip): per launch `rows` distinct random rows; the saturating launches
# have 26 * 262144 rows, the small ones 26 * 4096 (told apart by the grid: the larger grid is the saturating launch)
ROWS_SAT, ROWS_SMALL = 26 * 262144, 26 * 4096
for counter_tag in sorted(set(k.split("@")[0] for k in list(raw["FETCH_SIZE"]) + list(raw["WRITE_SIZE"]) if k.startswith("calib_rows_"))):
    nbytes = int(counter_tag.split("_")[2][:-1])
    rmw = "_rmw" in counter_tag
    x2 = counter_tag.endswith("_x2arrays")
    for rows, label in ((ROWS_SAT, "sat"), (ROWS_SMALL, "b4096")):
        cands_f = sorted((int(k.split("@grid")[1]), v) for k, v in raw["FETCH_SIZE"].items() if k.split("@")[0] == counter_tag)
        cands_w = sorted((int(k.split("@grid")[1]), v) for k, v in raw["WRITE_SIZE"].items() if k.split("@")[0] == counter_tag)
        if not cands_f:
            continue
        f = (cands_f[-1] if label == "sat" else cands_f[0])[1]
        w = ((cands_w[-1] if label == "sat" else cands_w[0])[1]) if cands_w else None
        rd = rows * nbytes * (2 if x2 else 1) + rows * 4
        wr = rows * nbytes * (2 if x2 else 1) if rmw else 0
        fac["%s_%s" % (counter_tag, label)] = {
            "fetch_raw_bytes": f["mean_KiB"] * 1024, "fetch_known_bytes": rd, "fetch_factor": rd / (f["mean_KiB"] * 1024),
            "write_raw_bytes": w["mean_KiB"] * 1024 if w else None, "write_known_bytes": wr,
            "write_factor": (wr / (w["mean_KiB"] * 1024)) if (w and w["mean_KiB"] and wr) else None}
out["calibration"] = fac
ff_gather = (fac["calib_gather"]["fetch_factor"] or 1.0)
ff_stream = (fac["calib_stream"]["fetch_factor"] or 1.0)
wf = (fac["calib_stream"]["write_factor"] or 1.0)
rmw = fac.get("calib_rows_128B_rmw_sat") or {}
ff_rmw128 = rmw.get("fetch_factor") or ff_gather
wf_rmw128 = rmw.get("write_factor") or wf
out["factors_used"] = {"fetch_gather": ff_gather, "fetch_stream": ff_stream, "write": wf,
                       "fetch_rmw128": ff_rmw128, "write_rmw128": wf_rmw128,
                       "note": "update kernels: read-modify-write of random 128-byte lines -> the rmw128 factors; gather: 64-byte rows"}
out["kernels"] = {}
for key in sorted(set(list(raw["FETCH_SIZE"]) + list(raw["WRITE_SIZE"]))):
    if key.startswith("calib"):
        continue
    f = raw["FETCH_SIZE"].get(key)
    w = raw["WRITE_SIZE"].get(key)
    fb = f["mean_KiB"] * 1024 if f else None
    wb = w["mean_KiB"] * 1024 if w else None
    upd = key.startswith("embed_update")
    out["kernels"][key] = {"fetch_raw_bytes": fb, "write_raw_bytes": wb,
                           "calibrated_bytes": ((fb or 0) * (ff_rmw128 if upd else ff_gather) + (wb or 0) * (wf_rmw128 if upd else wf)),
                           "fetch_bytes_gather_corrected": fb * ff_gather if fb is not None else None,
                           "fetch_bytes_x2": fb * 2 if fb is not None else None,
                           "write_bytes_corrected": wb * wf if wb is not None else None,
                           "launches": (f or w)["launches"]}
out["raw"] = raw
print(json.dumps(out, indent=1))
