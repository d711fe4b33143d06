import glob, json, sys
for p in sorted(glob.glob(sys.argv[1] + "/trace_*.json")):
    d = json.load(open(p))
    b = p.replace("trace_", "bench_")
    try:
        ms = json.load(open(b))["ms_per_step"]
    except Exception:
        ms = None
    t = d["train"]
    print(p.split("trace_")[-1][:-5], "ms/step", ms and round(ms, 4), " ".join("%s=%.2f" % (k.replace("fwd_", "f").replace("bwd_", "b").replace("layer", "L"), v["mean"]) for k, v in t.items() if isinstance(v, dict) and "barrier" not in k))
