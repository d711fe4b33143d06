#!/usr/bin/env python
"""The update's id-only pre-pass (dctr_embed_segments) at large batches: the two-level path (k_prepass_bin + k_prepass_sort)
against the paths it replaces (tag scan up to B = 32 768, k_bucket + bucket sort above), event-timed on rotating batches,
with the buckets compared bit for bit.  Diag library: DCTR_PREPASS_TWO_LEVEL_MIN moves the switch-over.
    python tools/prepass_bench.py [B ...]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch._hip import lib as L  # noqa: E402

L.use_diag_library()
import bench  # noqa: E402
from deepctr_torch._hip.ops import _ptr  # noqa: E402

Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384, 32768, 65536, 262144]
zipf = "--zipf" in sys.argv
sys.argv = ["bench.py"]
args = bench.parse()
dev = "cuda:0"
torch.cuda.set_device(0)
model = bench.build_model(args, dev)
plan = model.model_plan()
cplan = plan.bind(dev)
lib = L.lib()
s = L.stream_handle(dev)
nu = len(plan.units)
for B in Bs:
    ring = 3
    gen = torch.Generator().manual_seed(B)
    if zipf:
        r = torch.arange(1, args.vocab + 1, dtype=torch.float64).pow(-1.05)
        ids = torch.multinomial(r, ring * B * 26, replacement=True, generator=gen).reshape(ring * B, 26)
    else:
        ids = torch.randint(0, args.vocab, (ring * B, 26), generator=gen)
    X = torch.cat([ids.float(), torch.rand(ring * B, 13, generator=gen)], 1).to(dev)
    slots = []
    for j in range(ring):
        ids_t = torch.empty(nu, B, dtype=torch.int32, device=dev)
        parts_t = torch.empty(nu, B, dtype=torch.int16, device=dev)
        Xb = X[j * B:(j + 1) * B]
        L.check(lib.dctr_embed_ids(cplan, plan.units_ptr(), nu, _ptr(Xb), Xb.stride(0), B, _ptr(ids_t), _ptr(parts_t), s))
        slots.append((ids_t, parts_t))
    P = int(lib.dctr_embed_update_partitions(cplan, B))
    n_ws = int(lib.dctr_embed_update_workspace_ints(cplan, nu, B))
    res = {"B": B, "P": P, "ids": "zipf" if zipf else "uniform", "workspace_MB": n_ws * 4 / 1e6}
    snaps = {}
    for name, env in (("old", str(1 << 30)), ("two_level", "0")):
        os.environ["DCTR_PREPASS_TWO_LEVEL_MIN"] = env
        ws = torch.zeros(n_ws, dtype=torch.int32, device=dev)

        def run(j):
            ws[:nu * P].zero_()          # (the old bucket path counts with atomics: counters start at zero)
            L.check(lib.dctr_embed_segments(cplan, plan.units_ptr(), nu, plan.max_vocab, _ptr(slots[j % ring][0]),
                                            _ptr(slots[j % ring][1]), B, _ptr(ws), n_ws, s))
        for j in range(ring):
            run(j)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for j, (a, b) in enumerate(evs):
            ws[:nu * P].zero_()
            a.record()
            L.check(lib.dctr_embed_segments(cplan, plan.units_ptr(), nu, plan.max_vocab, _ptr(slots[j % ring][0]),
                                            _ptr(slots[j % ring][1]), B, _ptr(ws), n_ws, s))
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res[name + "_us"] = {"median": ts[len(ts) // 2], "min": ts[0]}
        run(0)
        torch.cuda.synchronize()
        cnt = ws[:nu * P].clone()
        keys = ws[nu * P:nu * P * 513].view(nu * P, 512).clone()
        snaps[name] = (cnt, keys)
    c0, k0 = snaps["old"]
    c1, k1 = snaps["two_level"]
    same_cnt = bool(torch.equal(c0, c1))
    ar = torch.arange(512, device=dev).unsqueeze(0)
    valid = (ar < c0.clamp(max=512).unsqueeze(1)) & (c0 <= 512).unsqueeze(1)
    same_keys = bool(torch.equal(k0[valid], k1[valid]))
    res.update(counts_equal=same_cnt, keys_equal=same_keys, entries=int(c0.sum().item()), max_count=int(c0.max().item()),
               overflowing_partitions=int((c0 > 512).sum().item()))
    print(json.dumps(res), flush=True)
    del X, slots
    torch.cuda.empty_cache()
