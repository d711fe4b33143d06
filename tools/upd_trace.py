#!/usr/bin/env python
"""Where does k_embed_update (the general update kernel; the pre-sorted path: tools/zipf_update_probe.py) spend its time?  Runs the DeepFM bench model's update kernel on ROTATING batches (so
table rows come from HBM, not from the 256 MB Infinity Cache like a same-batch timing loop) with per-workgroup
phase timestamps, and prints phase statistics + event-timed durations for a few partition counts.
    python tools/upd_trace.py [--opt adagrad] > gpurun_out/upd_trace.json"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch._hip import lib as _L  # noqa: E402

_L.use_diag_library()   # make -C deepctr-torch_amd/csrc diag

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", default="adagrad")
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    args = argparse.Namespace(batch=a.batch, vocab=1_000_000, optimizer=a.opt, ids="uniform")
    dev = "cuda:0"
    model = bench.build_model(args, dev)
    X, y = bench.synth(args, dev, 0)
    from deepctr_torch._hip import lib as L
    from deepctr_torch._hip.ops import _ptr
    lib = L.lib()
    B = a.batch
    plan = model.model_plan()
    model._train_step(X[:B], y[:B])          # binds plan, optimizer state
    cplan = plan.bind(dev)
    s = L.stream_handle(dev)
    nb = X.shape[0] // B
    out = torch.empty(B, plan.ld_out, device=dev)
    wide, fm = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fm_s = torch.empty(B, 16, device=dev)
    ids = [torch.empty(len(plan.units), B, dtype=torch.int32, device=dev) for _ in range(nb)]
    parts = [torch.empty(len(plan.units), B, dtype=torch.int16, device=dev) for _ in range(nb)]
    outs = []
    for j in range(nb):     # forward once per batch to get ids_t / parts_t / out / fm_s (kept per batch)
        o = torch.empty(B, plan.ld_out, device=dev)
        fs = torch.empty(B, 16, device=dev)
        L.check(lib.dctr_embed_fwd(cplan, _ptr(X[j * B:]), X.stride(0), B, _ptr(o), plan.ld_out, _ptr(wide), 1, _ptr(fm),
                                   None, plan.units_ptr(), len(plan.units), _ptr(ids[j]), _ptr(parts[j]), _ptr(fs), 16,
                                   s))
        outs.append((o, fs))
    g_out = torch.randn(B, plan.ld_out, device=dev) * 1e-3
    g_fm, g_wide = torch.randn(B, device=dev) * 1e-3, torch.randn(B, device=dev) * 1e-3
    lr = float(plan.update[1])
    eps = float(plan.update[2]) if a.opt == "adagrad" else 0.0
    code = L.UPD_ADAGRAD if a.opt == "adagrad" else L.UPD_SGD

    def upd(j):     # the GENERAL kernel (every workgroup scans and sorts for itself): the one that carries the stamps
        o, fs = outs[j]
        L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), len(plan.units), plan.max_vocab, _ptr(ids[j]),
                                      _ptr(parts[j]), B, _ptr(g_out), plan.ld_out, _ptr(o), plan.ld_out, _ptr(fs), 16,
                                      _ptr(g_fm), _ptr(g_wide), 1, code, lr, eps, None, 0, None, None, None, 0, 0, s))

    def fwd(j):
        L.check(lib.dctr_embed_fwd(cplan, _ptr(X[j * B:]), X.stride(0), B, _ptr(out), plan.ld_out, _ptr(wide), 1, _ptr(fm),
                                   None, plan.units_ptr(), len(plan.units), _ptr(ids[j]), _ptr(parts[j]), _ptr(fm_s), 16,
                                   s))

    def timed(fn, rot, n=48):
        for j in range(4):
            fn(j % nb if rot else 0)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n):
            ev[i][0].record()
            fn((i + 4) % nb if rot else 0)
            ev[i][1].record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return {"avg_us": sum(ts) / n, "min_us": ts[0], "med_us": ts[n // 2], "max_us": ts[-1]}

    res = {"B": B, "opt": a.opt, "fwd_same_batch": timed(fwd, False), "fwd_rotating": timed(fwd, True)}
    for lp in (-1, 32, 43, 64, 128):      # partitions per unit (-1: the library's choice)
        lib.dctr_dbg_update_trace(None, lp)
        res["upd_P_%d_same_batch" % lp] = timed(upd, False)
        res["upd_P_%d_rotating" % lp] = timed(upd, True)
    # phase trace at the default partitioning, rotating batches
    for lp in (-1, 64):
        nwg = 26 * ((B + 95) // 96 if lp < 0 else lp) + 16
        buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
        lib.dctr_dbg_update_trace(ctypes.c_void_p(buf.data_ptr()), lp)
        for j in range(8, 12):
            buf.zero_()
            upd(j)
        torch.cuda.synchronize()
        lib.dctr_dbg_update_trace(None, -1)
        t = buf.view(nwg, 8).cpu().numpy().astype("int64")
        live = t[t[:, 0] > 0]
        t0 = live[:, 0].min()
        tick = 10.0 / 1000.0   # wall_clock64: 100 MHz -> 0.01 us per tick
        work = live[live[:, 6] > 0]
        def st(v):
            import numpy as np
            v = np.asarray(v, dtype="float64") * tick
            return {"mean": float(v.mean()), "p10": float(np.percentile(v, 10)), "p50": float(np.percentile(v, 50)),
                    "p90": float(np.percentile(v, 90)), "max": float(v.max())}
        res["trace_P_%d" % lp] = {
            "n_wg": int(len(live)), "n_working": int(len(work)),
            "entries": st(work[:, 7] / tick),
            "start_after_first": st(work[:, 0] - t0),
            "scan": st(work[:, 1] - work[:, 0]), "sort": st(work[:, 2] - work[:, 1]),
            "issue_loads": st(work[:, 3] - work[:, 2]), "loads_land": st(work[:, 4] - work[:, 3]),
            "tile0_rest": st(work[:, 5] - work[:, 4]), "other_tiles": st(work[:, 6] - work[:, 5]),
            "wg_total": st(work[:, 6] - work[:, 0]), "end_after_first": st(work[:, 6] - t0),
        }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
