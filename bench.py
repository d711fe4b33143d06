#!/usr/bin/env python
"""bench.py -- training samples/sec of the DeepCTR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward + BCE(sum) + backward (fused sparse embedding update) + dense optimizer step of DeepFM on
one batch of 4096 synthetic Criteo-shaped samples (26 sparse x 1M-row vocab, 13 dense, emb_dim 16) whose
dataset (4096 x 64 rows) is already resident in HBM.  Prints ONE JSON line (rank 0).  Besides the
contract's keys it carries
  roofline     the embedding update (the step's HBM-bound kernel): ALGORITHMIC bytes per launch / its
               duration INSIDE the step (HIP events on its queue); scalar keys beside it name the step's longest launch
               (dominant_kernel / _bound / _avg_us / _frac, timed inside hipGraph replays through dctr_stamp) and the clock
               and power the box held (sclk_mhz, power_w)
  cpu_baseline the UNMODIFIED reference (unpacked from oracle/_ref/, kind "reference"), timed on this box's host cores at
               the best thread count of a sweep, on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))

import torch  # noqa: E402

F_SPARSE, N_DENSE, DIM = 26, 13, 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="deepfm", choices=["deepfm"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--optimizer", default="adagrad", choices=["adagrad", "sgd"])
    ap.add_argument("--ids", default="uniform", choices=["uniform", "zipf"],
                    help="id distribution of the synthetic batches: uniform over the vocabulary (BASELINE's config: the worst "
                         "case for caches) or Zipf(1.05) -- a secondary, clearly labelled run (SURVEY 8(d)): hot ids, long "
                         "duplicate segments in the update")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replays")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="train steps captured per hipGraph (the ~15 us launch gap is paid once per graph); 0 = the "
                         "largest divisor of --steps in [8, 32], else 16 plus a tail graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the xDeepFM / FiBiNET legs (BASELINE.json configs 3-4) of the N=1 line")
    ap.add_argument("--force-parallel", action="store_true", help="use the data-parallel trainer even with 1 rank")
    ap.add_argument("--no-saturating", action="store_true", help="skip the saturating-launch leg (B_eff 262 144)")
    ap.add_argument("--legs", default="all",
                    help="comma-separated subset of the N = 1 line's other_configs to run (xdeepfm, fibinet, deepfm_varlen, "
                         "default_kwargs, fit_api, sharded_1rank, fit_api_sharded_1rank); 'all' by default")
    ap.add_argument("--exchange", default=os.environ.get("DCTR_SHARDED_EXCHANGE", "auto"),
                    choices=["auto", "rccl", "direct", "try-direct"],
                    help="how the table-sharded step exchanges rows / gradients / dense gradients between ranks: 'rccl' = "
                         "torch.distributed collectives issued by the host between hipGraph segments; 'direct' = copies into "
                         "the peers' IPC-mapped buffers + arrival words, the whole step one hipGraph (validated with N "
                         "processes on ONE GPU and at one rank; never run across GPUs: no multi-GPU box was available to the "
                         "build); 'auto' (= 'try-direct') = direct at one rank; at N > 1 direct when the exchange's "
                         "self-test (bytes pushed into / pulled out of every peer's buffers by kernels, three rounds, "
                         "compared element by element) passes on EVERY rank, else rccl -- the JSON line and stderr say "
                         "which ran")
    ap.add_argument("--kernel-iters", type=int, default=50, help="event-timed launches per hot-path kernel")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed block of --steps steps is run this many times (each bracketed by a synchronize, and a "
                         "barrier with N > 1); ms_per_step / value are the MEDIAN block, all blocks are listed")
    ap.add_argument("--warmup-seconds", type=float, default=1.0,
                    help="besides --warmup steps: keep replaying until this much wall time has passed (clocks, caches and "
                         "the allocator settle; a 20-step block is 2 ms of GPU time)")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: the multi-process orchestration only (gloo instead of RCCL) -- there is NO CPU compute path: "
                         "it runs only where a test has put stand-ins under the C-ABI (tests/test_bench_gloo.py)")
    ap.add_argument("--diag-trace", default="",
                    help="development: load libdctr_hip_diag.so and save the tower kernels' per-workgroup phase stamps of "
                         "the last timed step (as replayed from the hipGraph) to this .npy file (tools/tower_trace.py)")
    return ap.parse_args()


def build_model(args, device):
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("C%d" % (i + 1), args.vocab, DIM) for i in range(F_SPARSE)] + \
           [DenseFeat("I%d" % (i + 1), 1) for i in range(N_DENSE)]
    model = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0,
                   seed=1024, device=device)
    model.compile(args.optimizer, "binary_crossentropy", metrics=[])
    model.train()
    return model


def synth(args, device, rank):
    gen = torch.Generator().manual_seed(rank)
    n = args.batch * 64
    if args.ids == "zipf":
        # Zipf(alpha = 1.05) over the vocabulary by inverse-CDF sampling of the continuous approximation
        a = 1.05
        u = torch.rand((n, F_SPARSE), generator=gen, dtype=torch.float64)
        vmax = float(args.vocab)
        ids = (((vmax ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))).floor().clamp_(1, vmax).long() - 1
    else:
        ids = torch.randint(0, args.vocab, (n, F_SPARSE), generator=gen)       # uniform: worst case for caches
    X = torch.cat([ids.float(), torch.rand(n, N_DENSE, generator=gen)], dim=1).to(device)
    y = torch.randint(0, 2, (n,), generator=gen).float().to(device)
    return X, y


def algorithmic_bytes(B, opt):
    """Per-launch algorithmic HBM bytes of the hand-written kernels."""
    ld = (F_SPARSE * DIM + N_DENSE + 3) // 4 * 4
    x_row = (F_SPARSE + N_DENSE) * 4
    rows, wrows = F_SPARSE * DIM * 4, F_SPARSE * 4
    side = F_SPARSE * 4 + DIM * 4                                              # ids_t + fm_s side outputs
    side += F_SPARSE * 2                                                       # + the 16-bit partition tags
    fwd = B * (x_row + rows + wrows + ld * 4 + 8 + side)
    n_rw = 4 if opt == "adagrad" else 2                                        # table (+state): read + write
    # ids_t + g_out + fm_s + g_fm + g_wide, then the row read-modify-writes (FM's backward is folded algebraically:
    # the forward's copy of the rows is not re-read)
    upd = B * (F_SPARSE * 4 + rows + DIM * 4 + 8 + n_rw * (rows + wrows))
    seg = B * (F_SPARSE * 2 + F_SPARSE * 4 + F_SPARSE * 4)   # partition tags, the ids of the entries kept, sorted keys out
    return {"embed_fwd": fwd, "embed_segments": seg, "embed_update": upd}


def time_hot_kernels(model, X_all, B, iters, opt, ring=16):
    """HIP-event timing of each hand-written embedding kernel on torch's current stream (the stream they launch
    on).  The launches ROTATE over `ring` different batches: re-timing one batch would serve every table row from
    the 256 MB Infinity Cache and overstate the HBM rate."""
    from deepctr_torch._hip import lib as L
    from deepctr_torch._hip.ops import _ptr
    lib = L.lib()
    plan = model.model_plan()
    dev = X_all.device
    ring = max(1, min(ring, X_all.shape[0] // B))
    wide, fm = torch.empty(B, device=dev), torch.empty(B, device=dev)
    g_out = torch.randn(B, plan.ld_out, device=dev) * 1e-3
    g_fm, g_wide = torch.randn(B, device=dev) * 1e-3, torch.randn(B, device=dev) * 1e-3
    cplan = plan.bind(dev)
    assert plan.update_kernel_ok(B), "bench shape must take the deterministic update kernel"
    s = L.stream_handle(dev)
    lr = float(plan.update[1])
    eps = float(plan.update[2]) if opt == "adagrad" else 0.0
    slots = [(X_all[j * B:(j + 1) * B], torch.empty(B, plan.ld_out, device=dev),
              torch.empty(len(plan.units), B, dtype=torch.int32, device=dev), torch.empty(B, DIM, device=dev),
              torch.empty(len(plan.units), B, dtype=torch.int16, device=dev))
             for j in range(ring)]

    def fwd(j):
        Xb, out, ids_t, fm_s, parts_t = slots[j % ring]
        L.check(lib.dctr_embed_fwd(cplan, _ptr(Xb), Xb.stride(0), B, _ptr(out), plan.ld_out, _ptr(wide), 1, _ptr(fm),
                                   None, plan.units_ptr(), len(plan.units), _ptr(ids_t), _ptr(parts_t), _ptr(fm_s),
                                   DIM, s))

    ws, ws_n = plan.update_workspace(B, dev, always=True)

    def seg(j):     # the id-only pre-pass of the update (launched right behind the forward in a train step)
        Xb, out, ids_t, fm_s, parts_t = slots[j % ring]
        L.check(lib.dctr_embed_segments(cplan, plan.units_ptr(), len(plan.units), plan.max_vocab, _ptr(ids_t),
                                        _ptr(parts_t), B, _ptr(ws), ws_n, s))

    def upd(j):
        Xb, out, ids_t, fm_s, parts_t = slots[j % ring]
        L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), len(plan.units), plan.max_vocab, _ptr(ids_t),
                                      _ptr(parts_t), B,
                                      _ptr(g_out), plan.ld_out, _ptr(out), plan.ld_out, _ptr(fm_s), DIM, _ptr(g_fm),
                                      _ptr(g_wide), 1, L.UPD_ADAGRAD if opt == "adagrad" else L.UPD_SGD, lr, eps,
                                      None, 0, None, None, _ptr(ws), ws_n, 1, s))

    stages = [("embed_fwd", fwd), ("embed_segments", seg), ("embed_update", upd)]
    for j in range(ring):       # every slot's side outputs exist before any update is timed
        fwd(j)
    for j in range(ring):
        seg(j)
        upd(j)
    torch.cuda.synchronize()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in stages]
          for _ in range(iters)]
    for it in range(iters):
        for k, (_, fn) in enumerate(stages):
            ev[it][k][0].record()
            fn(it)
            ev[it][k][1].record()
    torch.cuda.synchronize()
    res = {}
    for k, (name, _) in enumerate(stages):
        ts = sorted(ev[it][k][0].elapsed_time(ev[it][k][1]) * 1e3 for it in range(iters))  # us
        res[name] = {"avg_us": sum(ts) / len(ts), "min_us": ts[0], "median_us": ts[len(ts) // 2]}
    return res


def time_update_in_step(model, X, y, B, n=80):
    """The embedding update's duration INSIDE the train step (round-3 verdict: the stand-alone event timing of
    time_hot_kernels is ~13 % kinder than what the kernel does beside the weight-gradient kernels it shares the chip with):
    HIP events recorded on the update's own queue around its launch, in eager two-queue steps of the real step engine
    (deepctr_torch/_hip/step.py) -- same kernels, same overlap as the graph-replayed step.  None when the model does not
    run on the engine."""
    st = model._fused_step_state()
    eng = st.get("engine") if st else None
    if eng is None:
        return None
    nb = X.shape[0] // B
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    evt = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    try:
        for i in range(n):
            eng.timing = evs[i]
            eng.timing_tower = evt[i]
            j = i % nb
            model._train_step(X[j * B:(j + 1) * B], y[j * B:(j + 1) * B])
    finally:
        eng.timing = None
        eng.timing_tower = None
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[n // 4:])      # us; the first quarter warms up
    tt = sorted(a.elapsed_time(b) * 1e3 for a, b in evt[n // 4:])
    return {"avg_us": sum(ts) / len(ts), "median_us": ts[len(ts) // 2], "min_us": ts[0], "launches": len(ts),
            "how": "HIP events on the update's queue around dctr_embed_update in eager two-queue train steps",
            "tower": {"avg_us": sum(tt) / len(tt), "median_us": tt[len(tt) // 2], "min_us": tt[0], "launches": len(tt),
                      "how": "HIP events on the main queue around dctr_embed_tower_train_step in the same eager steps"}}


def time_kernels_in_graph(model, X, y, B, S=20, replays=6):
    """The step's two long launches timed INSIDE hipGraph replays (round-5 verdict: eager-step HIP events put the tower launch
    at 57 us where the graph-replayed launch takes 49.5): a group of S steps is captured with one-thread dctr_stamp launches
    around the tower launch (main queue) and around the embedding update (its queue) that write the device's 100 MHz wall
    clock; the durations are differences of those stamps, averaged over every step of `replays` replays.  A stamp launch
    costs the queue ~1-2 us, which the figures include (rocprofv3's own kernel durations under profiles/ are the cross-check).
    None when the model does not run on the step engine."""
    st = model._fused_step_state()
    eng = st.get("engine") if st else None
    if eng is None:
        return None
    from deepctr_torch._hip.graph import GraphedTrainStep
    nb = X.shape[0] // B
    S = min(S, nb)
    stamps = torch.zeros((S, 5), dtype=torch.int64, device=X.device)
    try:
        eng.stamps, eng.stamp_i = stamps, 0
        g = GraphedTrainStep(model, X[:B], y[:B], steps_per_graph=S, double_buffer=False, inputs_ready=True).capture(X[:B], y[:B])
    finally:
        eng.stamps = None
    tower, upd, per, cost = [], [], [], []
    for r in range(replays):
        g.step_block(X[:S * B], y[:S * B])
        torch.cuda.synchronize()
        v = stamps.cpu().numpy().astype("float64") / 100.0          # us
        if r == 0:
            continue                                                  # (first replay: cold)
        tower += list(v[:, 1] - v[:, 0])
        upd += list(v[:, 3] - v[:, 2])
        per += list(v[1:, 0] - v[:-1, 0])
        cost += list(v[:, 0] - v[:, 4])
    del g
    mean = lambda a: float(sum(a) / len(a)) if a else None          # noqa: E731
    c = mean(cost) or 0.0
    return {"tower_us": mean(tower) - c, "update_us": mean(upd) - c, "stamp_launch_us": c, "tower_us_raw": mean(tower),
            "update_us_raw": mean(upd), "step_period_us": mean(per), "steps": len(tower),
            "how": "dctr_stamp launches (device wall clock, 10 ns ticks) around the launch inside %d-step hipGraph replays: "
                   "the time the launch holds its queue (its end-of-kernel write-back included, which rocprofv3's kernel "
                   "duration leaves out), less what one stamp launch costs (two stamps back to back); the stamps lengthen "
                   "the step itself by ~8 %%" % S}


def gpu_clock_power(busy):
    """(sclk MHz, power W) read from rocm-smi while `busy()` keeps the GPU replaying train steps -- so that a number measured
    on a box that clocks lower can be told from a slower kernel (round-5 verdict: 11-16 % box-to-box spread on the MFMA-bound
    legs).  None where rocm-smi does not answer."""
    import re
    import subprocess
    try:
        busy()
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        busy()
        torch.cuda.synchronize()
        d = json.loads(out[out.index("{"):])
        card = d[sorted(d)[0]]
        sclk = power = None
        for k, v in card.items():
            kl = k.lower()
            if sclk is None and "sclk" in kl:
                m = re.search(r"(\d+)\s*mhz", str(v).lower())
                if m:
                    sclk = float(m.group(1))
            if power is None and "power" in kl and "(w)" in kl:
                try:
                    power = float(v)
                except (TypeError, ValueError):
                    pass
        return {"sclk_mhz": sclk, "power_w": power}
    except Exception as exc:
        return {"sclk_mhz": None, "power_w": None, "error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}


def saturating_launch(args, model, device, B_sat=262144):
    """SURVEY.md 8(d) number (3): the embedding kernels at a saturating launch (B_eff = 262 144: every CU holds many
    workgroups, the request queues of the memory system stay full), rotating over two different batches."""
    gen = torch.Generator().manual_seed(12345)
    n = 2 * B_sat
    ids = torch.randint(0, args.vocab, (n, F_SPARSE), generator=gen)
    Xs = torch.cat([ids.float(), torch.rand(n, N_DENSE, generator=gen)], dim=1).to(device)
    k = time_hot_kernels(model, Xs, B_sat, 6, args.optimizer, ring=2)
    alg = algorithmic_bytes(B_sat, args.optimizer)
    out = {"B_eff": B_sat, "kernels": {}}
    for name, v in k.items():
        out["kernels"][name] = {"avg_us": v["avg_us"], "min_us": v["min_us"], "alg_bytes": alg[name],
                                "gbs": alg[name] / (v["avg_us"] * 1e-6) / 1e9,
                                "frac_of_hbm_peak": alg[name] / (v["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
    path_us = k["embed_segments"]["avg_us"] + k["embed_update"]["avg_us"]
    path_b = alg["embed_segments"] + alg["embed_update"]
    out["gather_frac_of_hbm_peak"] = out["kernels"]["embed_fwd"]["frac_of_hbm_peak"]
    out["update_path_frac_of_hbm_peak"] = path_b / (path_us * 1e-6) / 1e9 / HBM_PEAK_GBS
    out["update_path_us"] = path_us
    # the predict path (round 5): a model that was never compiled seats deep row + wide weight of an id in one 128-byte line
    # (_hip/layout.py apply_infer_layout); the same gather kernel without the update's side outputs, on that layout and on
    # the training layout
    try:
        from deepctr_torch._hip import lib as L
        from deepctr_torch._hip.layout import apply_infer_layout
        from deepctr_torch._hip.ops import _ptr
        from deepctr_torch.inputs import DenseFeat, SparseFeat
        from deepctr_torch.models import DeepFM
        cols = [SparseFeat("C%d" % (i + 1), args.vocab, DIM) for i in range(F_SPARSE)] + \
               [DenseFeat("I%d" % (i + 1), 1) for i in range(N_DENSE)]
        pm = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0, seed=1024,
                    device=device)
        pplan = pm.model_plan()
        seated = apply_infer_layout(pplan)
        lib = L.lib()
        ld = (F_SPARSE * DIM + N_DENSE + 3) // 4 * 4
        fwd_b = B_sat * ((F_SPARSE + N_DENSE) * 4 + F_SPARSE * DIM * 4 + F_SPARSE * 4 + ld * 4 + 8)
        res = {}
        for tag, plan_ in (("predict_layout", pplan), ("training_layout", model.model_plan())):
            cp = plan_.bind(Xs.device)
            o = torch.empty(B_sat, plan_.ld_out, device=Xs.device)
            w_, f_ = torch.empty(B_sat, device=Xs.device), torch.empty(B_sat, device=Xs.device)
            s_ = L.stream_handle(Xs.device)
            evs = []
            for it in range(8):
                Xb = Xs[(it % 2) * B_sat:(it % 2 + 1) * B_sat]
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                L.check(lib.dctr_embed_fwd(cp, _ptr(Xb), Xb.stride(0), B_sat, _ptr(o), plan_.ld_out, _ptr(w_), 1, _ptr(f_), None,
                                           plan_.units_ptr(), plan_.n_grid_units, None, None, None, 0, s_))
                b_.record()
                evs.append((a, b_))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in evs[2:])
            us = sum(ts) / len(ts)
            res[tag] = {"avg_us": us, "min_us": ts[0], "alg_bytes": fwd_b, "gbs": fwd_b / (us * 1e-6) / 1e9,
                        "frac_of_hbm_peak": fwd_b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
            del o, w_, f_
        res["units_seated"] = seated
        out["embed_fwd_forward_only"] = res
        del pm
    except Exception as exc:
        out["embed_fwd_forward_only"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    del Xs
    torch.cuda.empty_cache()
    return out


PMC_SOURCES = ("deepctr-torch_amd/csrc/update.hip", "deepctr-torch_amd/csrc/update_kernels.hpp",
               "deepctr-torch_amd/csrc/update_launch.inc", "deepctr-torch_amd/csrc/embed.hip",
               "deepctr-torch_amd/csrc/common.hpp", "deepctr-torch_amd/csrc/lazy_opt.hpp")


def kernel_code_hash():
    import hashlib
    h = hashlib.sha256()
    for rel in PMC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(kernel, opt, B):
    """HBM bytes per launch of a hand-written kernel from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs, KiB units; a PMC pass cannot run inside this process).  The
    summary carries the sha256 of the kernel sources it was taken on (tools/pmc_summary.py): a summary of OTHER code is
    refused -- `traffic` is then null and `traffic_detail.stale` says why -- instead of silently describing a kernel that
    no longer exists.  `bytes` = raw counters x the factors calibrated on the kernel's own access pattern (random
    128-byte lines read-modify-written for the update, 64-byte rows for the gather: tools/micro/rowbench.hip under the same
    counters); `raw` is the uncorrected sum."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not paths:
        return None
    with open(paths[-1]) as fh:
        d = json.load(fh)
    src = "profiles/" + os.path.basename(paths[-1])
    if d.get("code_hash") != kernel_code_hash():
        return {"bytes": None, "stale": "%s was measured on other kernel sources (code_hash mismatch): rerun "
                                         "tools/pmc_traffic.sh" % src, "source": src}
    tag = "embed_fwd" if kernel == "embed_fwd" else "embed_update_%s" % opt
    # the PMC driver launches every kernel at B = 4096 and B = 32768: the smaller grid is this bench's launch
    cands = sorted((int(k.split("@grid")[1]), v) for k, v in d.get("kernels", {}).items() if k.split("@grid")[0] == tag)
    if not cands or B != 4096:
        return None
    best = cands[0][1]
    raw = best["fetch_raw_bytes"] + best["write_raw_bytes"]
    cal = d.get("calibration", {})
    f128 = (cal.get("calib_rows_128B_rmw_b4096") or {}).get("fetch_factor") or 2.0     # measured: 2.00
    f64 = (cal.get("calib_rows_64B_read_b4096") or {}).get("fetch_factor") or 1.0      # measured: 1.03
    # FETCH_SIZE tallies a 128-byte request at 64 bytes (factor 2.00 on random 128-byte lines, read or read-modify-write)
    # and smaller requests in full (1.03 on random 64-byte rows); WRITE_SIZE needs no correction (1.00).  The update reads
    # ONE 128-byte line per entry (row + Adagrad state; plain 64-byte rows under SGD), everything else in <= 64-byte
    # pieces: its fetch is the raw count + 64 bytes per such line.  The gather reads 64-byte rows only.
    lines128 = F_SPARSE * B if (kernel != "embed_fwd" and opt == "adagrad") else 0
    fetch = best["fetch_raw_bytes"] * f64 + lines128 * 64 * (f128 - 1.0)
    return {"bytes": fetch + best["write_raw_bytes"], "raw": raw, "upper_bound_all_fetches_x2": best.get("calibrated_bytes"),
            "fetch_raw": best["fetch_raw_bytes"], "write_raw": best["write_raw_bytes"], "lines_128B": lines128,
            "factor_128B": f128, "factor_64B": f64, "code_hash": d.get("code_hash"), "source": src}


def reference_cpu_baseline(args):
    """The UNMODIFIED reference on this box's host cores: `__graft_entry__.build()`
    leaves an archive of the reference package under the git-ignored oracle/_ref/ (it travels to the GPU box with the
    snapshot, like the built .so files; /root/reference itself does not exist there) and oracle/time_reference.py
    times its inner train step exactly as basemodel.py:242-262, one fresh process per variant.  None when the copy is
    absent (the port of oracle/torch_port.py is timed then)."""
    import subprocess
    import tarfile
    import tempfile
    arc = os.path.join(ROOT, "oracle", "_ref", "reference_deepctr_torch.tar.gz")
    if not os.path.isfile(arc):
        return None
    ref_root = tempfile.mkdtemp(prefix="dctr_ref_")
    with tarfile.open(arc) as tf:
        tf.extractall(ref_root)
    def run(extra, steps, threads):
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "time_reference.py"), "--reference-root", ref_root,
               "--batch", str(args.batch), "--vocab", str(args.vocab), "--steps", str(steps), "--threads", str(threads),
               "--json"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    # Thread sweep (round-4 verdict: torch's default of one thread per logical cpu OVERSUBSCRIBES this workload -- 128
    # threads: 1 646 samples/s; the survey's 8 threads: 4 781): the like-for-like variant at 8 ... ncpu threads, 3 timed
    # steps each; the best count then runs every variant with the full step count.  `value` is the BEST the host gives.
    ncpu = os.cpu_count() or 1
    like_extra = ["--optimizer", args.optimizer, "--l2", "0"]
    sweep = {}
    try:
        for th in [t for t in (8, 16, 32, 64, 128) if t <= ncpu] or [ncpu]:
            sweep[th] = run(like_extra, 3, th)["value"]
        best = max(sweep, key=sweep.get)
    except Exception as exc:
        print("reference cpu_baseline thread sweep failed: %s" % exc, file=sys.stderr)
        return None
    variants = {}
    for tag, extra, steps in (("like_for_like_l2_0_%s" % args.optimizer, like_extra, args.cpu_steps),
                              ("forward_only", ["--forward-only"], 4 * args.cpu_steps),
                              ("reference_defaults_l2_1e-5_adam", ["--optimizer", "adam", "--l2", "1e-5"],
                               max(2, args.cpu_steps // 3))):
        try:
            variants[tag] = run(extra, steps, best)
        except Exception as exc:
            print("reference cpu_baseline leg %s failed: %s" % (tag, exc), file=sys.stderr)
            return None
    like = variants["like_for_like_l2_0_%s" % args.optimizer]
    return {"value": like["value"], "unit": "samples/s", "cores": like.get("threads", best), "threads": best,
            "thread_sweep_samples_per_s": {str(k): v for k, v in sorted(sweep.items())},
            "kind": "reference",
            "sample": "%d train steps (after 2 warm-up) of the UNMODIFIED reference (unpacked from oracle/_ref/, timed as "
                      "basemodel.py:242-262) on the same DeepFM/batch=%d/vocab=%d workload, %s, l2=0, at the best thread "
                      "count of a sweep (%d of %d logical cpus)" % (like["steps"], args.batch, args.vocab, args.optimizer,
                                                                     best, ncpu),
            "ms_per_step": like["ms_per_step"], "variants": variants}


def cpu_baseline(args):
    """The reference's dense-gradient algorithm (torch-CPU port, oracle/torch_port.py: the same ATen calls in the same
    order; 1.1x the real reference's step time on the build container, profiles/r03_reference_cpu_timing.json) on this
    box's host cores, bounded samples of the bench workload.  SURVEY.md 8(d)'s three variants: (i) the reference's defaults
    (l2 = 1e-5 on every table and on Linear, Adam), (ii) like for like with the GPU line (l2 = 0, the bench's optimizer)
    -- this one is `value` --, (iii) forward only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    ref = reference_cpu_baseline(args)
    if ref is not None:
        return ref
    from torch_port import DeepFMPort, make_optimizer, train_step
    gen = torch.Generator().manual_seed(0)
    X = torch.cat([torch.randint(0, args.vocab, (args.batch, F_SPARSE), generator=gen).float(),
                   torch.rand(args.batch, N_DENSE, generator=gen)], dim=1)
    y = torch.randint(0, 2, (args.batch,), generator=gen).float()

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n
        return {"ms_per_step": dt * 1e3, "value": args.batch / dt, "unit": "samples/s", "steps": n, "warmup": warm}

    variants = {}
    torch.manual_seed(0)
    port = DeepFMPort(F_SPARSE, args.vocab, DIM, N_DENSE, hidden=(256, 128))
    opt = make_optimizer(port, args.optimizer)
    like = timed(lambda: train_step(port, opt, X, y), args.cpu_steps, 2)
    variants["like_for_like_l2_0_%s" % args.optimizer] = like
    port.eval()
    with torch.no_grad():
        variants["forward_only"] = timed(lambda: port(X), 4 * args.cpu_steps, 3)
    port.train()
    del opt
    adam = make_optimizer(port, "adam")
    variants["reference_defaults_l2_1e-5_adam"] = timed(lambda: train_step(port, adam, X, y, 1e-5, 1e-5),
                                                        max(2, args.cpu_steps // 3), 1)
    return {"value": like["value"], "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps (after 2 warm-up) of the same DeepFM/batch=%d/vocab=%d workload, dense [V,D] "
                      "gradients + dense torch.optim.%s like the reference, l2=0; host has %d logical cpus" % (
                          args.cpu_steps, args.batch, args.vocab, args.optimizer, os.cpu_count() or 0),
            "ms_per_step": like["ms_per_step"], "variants": variants}


class StepRunner(object):
    """K train steps with NO eager step inside the timed region: whole groups of ``S`` steps replay the main hipGraph,
    the K mod S leftover steps replay a second, shorter graph captured beforehand (round 1 timed 4 of 20 driver steps
    eagerly at ~3x the replayed cost)."""

    def __init__(self, model, X, y, B, steps, S, use_graph):
        self.model, self.X, self.y, self.B = model, X, y, B
        self.n_batches = X.shape[0] // B
        self.i = 0
        self.main = self.tail = None
        self.S = max(1, min(int(S), steps)) if use_graph else 0
        self.tail_n = (steps % self.S) if self.S else 0
        self.use_graph = use_graph

    def batch(self, i):
        j = i % self.n_batches
        return self.X[j * self.B:(j + 1) * self.B], self.y[j * self.B:(j + 1) * self.B]

    def eager(self, n):
        out = None
        for _ in range(n):
            out = self.model._train_step(*self.batch(self.i))
            self.i += 1
        return out

    def capture(self):
        from deepctr_torch._hip.graph import GraphedTrainStep
        xb, yb = self.batch(self.i)
        # (the synthetic dataset was uploaded and synchronised long before: its slices are complete)
        self.main = GraphedTrainStep(self.model, xb, yb, steps_per_graph=self.S, inputs_ready=True).capture(xb, yb)
        if self.tail_n:
            self.tail = GraphedTrainStep(self.model, xb, yb, steps_per_graph=self.tail_n, double_buffer=False,
                                         inputs_ready=True).capture(xb, yb)

    def _group(self, g, n):
        j = self.i % self.n_batches
        if j + n <= self.n_batches:      # consecutive resident rows: two staging copies for the whole group
            out = g.step_block(self.X[j * self.B:(j + n) * self.B], self.y[j * self.B:(j + n) * self.B])
            self.i += n
            return out
        out = None
        for _ in range(n):
            out = g(*self.batch(self.i))
            self.i += 1
        return out

    def run(self, steps):
        """`steps` train steps; every one of them a graph replay when graphs are on."""
        if self.main is None:
            return self.eager(steps)
        out, done = None, 0
        while steps - done >= self.S:
            out = self._group(self.main, self.S)
            done += self.S
        if steps - done:
            assert self.tail is not None and steps - done == self.tail_n, "tail graph does not match the step count"
            out = self._group(self.tail, self.tail_n)
        return out


def time_steps(model, X, y, B, steps, warmup, S, use_graph, repeats=1, warmup_seconds=0.0, sync=None):
    """Warm up (>= `warmup` steps: 3 eager ones, then replays of every captured graph until `warmup_seconds` have passed),
    then time `repeats` blocks of exactly `steps` steps, each bracketed by `sync()` (default: torch.cuda.synchronize; the
    multi-GPU caller adds its barrier).  Returns the MEDIAN block's elapsed time first, every block's in `times`."""
    sync = sync or torch.cuda.synchronize
    r = StepRunner(model, X, y, B, steps, S, use_graph)
    n_eager = min(3, max(1, warmup)) if use_graph else warmup
    r.eager(n_eager)
    graphed = False
    if use_graph:
        try:
            r.capture()
            graphed = True
        except Exception as exc:  # capture is an optimisation: report and continue eagerly
            print("hipGraph capture failed (%s: %s); running eager" % (type(exc).__name__, exc), file=sys.stderr)
            r.main = r.tail = None
            torch.cuda.synchronize()
    did = n_eager
    t_w = time.perf_counter()
    if graphed:
        while did < warmup or did == n_eager or time.perf_counter() - t_w < warmup_seconds:
            r.run(r.S + r.tail_n)        # one replay of the main graph (+ one of the tail graph)
            did += r.S + r.tail_n
            if warmup_seconds > 0:
                torch.cuda.synchronize()
    else:
        r.eager(max(0, warmup - n_eager))
        while time.perf_counter() - t_w < warmup_seconds:
            r.eager(steps)
            did += steps
            torch.cuda.synchronize()
    times, out = [], None
    for _ in range(max(1, repeats)):
        sync()
        t0 = time.perf_counter()
        out = r.run(steps)
        sync()
        times.append(time.perf_counter() - t0)
    elapsed = sorted(times)[len(times) // 2]
    return elapsed, out, graphed, did, r, times


def spread(times, steps):
    ms = sorted(t / steps * 1e3 for t in times)
    return {"blocks": len(ms), "ms_per_step_median": ms[len(ms) // 2], "ms_per_step_min": ms[0], "ms_per_step_max": ms[-1],
            "ms_per_step_all": [round(v, 5) for v in (t / steps * 1e3 for t in times)]}


# BASELINE.json configs[2] and configs[3]: same Criteo shape, other interaction layers
MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA peak
MFMA_MEASURED_TFLOPS = 133.0
OTHER = {
    "xdeepfm": dict(cls="xDeepFM", kwargs=dict(dnn_hidden_units=(256, 256), cin_layer_size=(128, 128), cin_split_half=True),
                    flop_per_sample=29.9e6, ref="xdeepfm.py:79-107, interaction.py:207-248",
                    workload="xDeepFM synthetic Criteo, CIN layers=[128,128] split_half, dnn=(256,256)"),
    "fibinet": dict(cls="FiBiNET", kwargs=dict(dnn_hidden_units=(128, 128)), flop_per_sample=9.1e6,
                    ref="fibinet.py:76-102, interaction.py:93-101,140-156",
                    workload="FiBiNET synthetic Criteo, SENET(reduction 3) + bilinear 'interaction', dnn=(128,128)"),
}


def build_other(name, args, device):
    from deepctr_torch import models as M
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    cols = [SparseFeat("C%d" % (i + 1), args.vocab, DIM) for i in range(F_SPARSE)] + \
           [DenseFeat("I%d" % (i + 1), 1) for i in range(N_DENSE)]
    spec = OTHER[name]
    model = getattr(M, spec["cls"])(cols, cols, l2_reg_linear=0, l2_reg_embedding=0, dnn_dropout=0, seed=1024,
                                    device=device, **spec["kwargs"])
    model.compile(args.optimizer, "binary_crossentropy", metrics=[])
    model.train()
    return model


def deepfm_leg(tag, args, device, X, y):
    """DeepFM legs beside the headline, same timed protocol (hipGraph replays, median block):
      default_kwargs  the reference's DEFAULT kwargs -- l2_reg_embedding = l2_reg_linear = 1e-5 (deepfm.py:41) and
                      compile('adam') (the reference's tests/utils.py:157) -- on the exact lazy update (csrc/lazy.hip)
      deepfm_varlen   the headline's 26 + 13 columns plus ONE pooled history (VarLenSparseFeat, maxlen 8, mean, ids != 0 mask,
                      its own 1M-row table on the deep and the wide side): the north_star's "EmbeddingBag backward"
    """
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_torch.models import DeepFM
    try:
        cols = [SparseFeat("C%d" % (i + 1), args.vocab, DIM) for i in range(F_SPARSE)]
        Xl, opt, l2 = X, args.optimizer, 0.0
        if tag == "default_kwargs":
            opt, l2 = "adam", 1e-5
            workload = "DeepFM synthetic Criteo, the reference's default kwargs: l2_reg_embedding = l2_reg_linear = 1e-5, adam"
        else:
            T = 8
            cols = cols + [VarLenSparseFeat(SparseFeat("hist", args.vocab, DIM), maxlen=T, combiner="mean")]
            gen = torch.Generator().manual_seed(77)
            n = X.shape[0]
            hist = torch.randint(1, args.vocab, (n, T), generator=gen)
            length = torch.randint(1, T + 1, (n, 1), generator=gen)
            hist = hist * (torch.arange(T).unsqueeze(0) < length)            # padded with id 0 (the mask, inputs.py:146)
            # column order of build_input_features (inputs.py:99-123): sparse, dense, then the VarLen positions
            Xl = torch.cat([X, hist.float().to(X.device)], dim=1).contiguous()
            workload = "DeepFM synthetic Criteo + one pooled history (VarLenSparseFeat maxlen 8, mean, mean length 4.5)"
        cols = cols[:F_SPARSE] + [DenseFeat("I%d" % (i + 1), 1) for i in range(N_DENSE)] + cols[F_SPARSE:]
        model = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=l2, l2_reg_embedding=l2, dnn_dropout=0,
                       seed=1024, device=device)
        model.compile(opt, "binary_crossentropy", metrics=[])
        model.train()
        elapsed, out, graphed, did, _, times = time_steps(model, Xl, y, args.batch, args.steps, args.warmup,
                                                          args.steps_per_graph, not args.no_graph, args.repeats,
                                                          args.warmup_seconds)
        model.model_plan().check_ids()
        st = model._fused_step_state()
        plan = model.model_plan()
        res = {"workload": workload + ", batch=%d, fwd+bwd+%s" % (args.batch, opt), "value": args.batch * args.steps / elapsed,
               "unit": "samples/s", "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps, "warmup": did,
               "hip_graph": graphed, "final_loss": float(out[0].item()), "timing": spread(times, args.steps),
               "update_mode": plan.update[0], "unit_path": bool(plan.unit_path),
               "step_engine": bool(st is not None and st.get("engine") is not None and st["engine"].steps_run > 0)}
        if tag == "default_kwargs" and plan.update[0] == "lazy":
            # The lazy update replays, for every row, the optimizer steps the row slept through (an L2 term / Adam's moments
            # move every row at every step: basemodel.py:412-428): per step it owes one optimizer step for EVERY row of every
            # table, paid when a batch draws the row, by the per-step sweep of one K-th of the rows, or by the next flush.
            # The shared synthetic set is 64 batches in a cycle (74 % of the rows are never drawn); the second measurement
            # draws 1024 distinct batches, where every row comes back after V / B = 244 steps on average.  With the sweep
            # (K = 32) both pay the whole debt inside the timed region; without it (DCTR_LAZY_SWEEP_K=0) the first left 74 %
            # of it to a flush that the timed region never ran (0.29-0.30 ms) and the second stalled on its longest
            # sleepers (1.2-1.5 ms).
            res["rows_repeat_every_steps"] = int(X.shape[0] // args.batch)
            try:
                nb = 1024
                gen = torch.Generator(device=device).manual_seed(99)
                Xs = torch.cat([torch.randint(0, args.vocab, (nb * args.batch, F_SPARSE), generator=gen, device=device).float(),
                                torch.rand((nb * args.batch, N_DENSE), generator=gen, device=device)], dim=1)
                ys = torch.randint(0, 2, (nb * args.batch,), generator=gen, device=device).float()
                el2, _, g2, did2, _, t2 = time_steps(model, Xs, ys, args.batch, args.steps, max(args.warmup, nb + 64),
                                                     args.steps_per_graph, not args.no_graph, args.repeats, 0.0)
                res["sweep_k"] = int(getattr(plan.lazy, "sweep_k", 0))
                res["steady_state"] = {"distinct_batches": nb, "mean_steps_between_draws": round(args.vocab / args.batch, 1),
                                       "ms_per_step": el2 / args.steps * 1e3, "value": args.batch * args.steps / el2,
                                       "unit": "samples/s", "warmup": did2, "hip_graph": g2,
                                       "timing": spread(t2, args.steps),
                                       "note": "uniform ids over fresh batches: every row is drawn again and again"}
                del Xs, ys
            except Exception as exc:
                torch.cuda.synchronize()
                res["steady_state"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
        del model, Xl
        torch.cuda.empty_cache()
        return res
    except Exception as exc:
        torch.cuda.synchronize()
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}


def sharded_1rank(args):
    """The table-sharded step (ShardedTrainer, direct exchange, S steps per hipGraph) at ONE rank, as a child process of
    this very script (`--force-parallel`): every exchange executes (posts and waits on this rank's own arrival words), so the
    line prices the sharded step's launches and boundaries while no multi-GPU node times the real thing."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-parallel", "--exchange", "direct",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-other-configs", "--no-cpu-baseline",
           "--batch", str(args.batch), "--vocab", str(args.vocab), "--optimizer", args.optimizer, "--kernel-iters", "2",
           "--no-saturating"]
    env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"workload": d["config"]["workload"], "parallelism": d["config"]["parallelism"], "value": d["value"],
                "unit": "samples/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "timing": d["timing"],
                "exchange": d["config"].get("exchange"), "steps_per_graph": d["config"].get("steps_per_graph"),
                "final_loss": d["final_loss"]}
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}


def fit_api_sharded_1rank(args):
    """``model.fit()`` through the table-sharded trainer at ONE rank (child process, DCTR_FIT_FORCE_TRAINER=1): the public call a
    torchrun user makes, with the exchange resolved by parallel.resolve_exchange ('auto') and full groups of steps replayed as
    one hipGraph each (distributed_fit.py) -- next to `sharded_1rank`, the trainer driven directly."""
    import subprocess
    code = (
        "import sys, os, json, time, io, contextlib\n"
        "sys.path.insert(0, os.path.join(%r, 'deepctr-torch_amd'))\n"
        "sys.argv = ['bench.py', '--batch', '%d', '--vocab', '%d', '--optimizer', '%s']\n"
        "import torch\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "args = b.parse()\n"
        "torch.cuda.set_device(0)\n"
        "model = b.build_model(args, 'cuda:0')\n"
        "X, y = b.synth(args, 'cuda:0', 0)\n"
        "X, y = X.repeat(4, 1), y.repeat(4)\n"
        "sink = io.StringIO()\n"
        "with contextlib.redirect_stdout(sink):\n"
        "    model.fit(X, y, batch_size=args.batch, epochs=2, verbose=0, shuffle=False)\n"    # (every block variant captured)
        "    torch.cuda.synchronize(); t0 = time.perf_counter()\n"
        "    h = model.fit(X, y, batch_size=args.batch, epochs=3, verbose=0, shuffle=False)\n"
        "    torch.cuda.synchronize(); dt = time.perf_counter() - t0\n"
        "n = X.shape[0]; steps = 3 * (n // args.batch)\n"
        "tr = model._dist_trainer\n"
        "print(json.dumps({'value': 3 * n / dt, 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'trainer': type(tr).__name__,\n"
        "                  'exchange': tr.exchange, 'exchange_note': getattr(tr.tr, 'exchange_note', None),\n"
        "                  'steps_per_graph': tr.blocks(X.device), 'last_epoch_loss': float(h.history['loss'][-1])}))\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), args.batch, args.vocab, args.optimizer, os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + os.getpid() % 300), WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", DCTR_FIT_FORCE_TRAINER="1")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        d.update(call="model.fit(x, y, batch_size=%d, epochs=3, shuffle=False) under DCTR_FIT_FORCE_TRAINER=1, WORLD_SIZE=1" % args.batch,
                 unit="samples/s")
        return d
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300]), "stderr": (r.stderr[-400:] if "r" in dir() else None)}


def other_config(name, args, device, X, y):
    """One more bench leg at N = 1: the same timed protocol on another model of BASELINE.json; MFMA-bound, so its
    roofline is algorithmic FLOP / (time x dense fp32 MFMA peak)."""
    spec = OTHER[name]
    try:
        model = build_other(name, args, device)
        elapsed, out, graphed, did, _, times = time_steps(model, X, y, args.batch, args.steps, args.warmup,
                                                          args.steps_per_graph, not args.no_graph, args.repeats,
                                                          args.warmup_seconds)
        model.model_plan().check_ids()
        ms = elapsed / args.steps * 1e3
        sps = args.batch * args.steps / elapsed
        tf = sps * spec["flop_per_sample"] / 1e12
        res = {"workload": spec["workload"] + ", batch=%d, fwd+bwd+%s, l2=0" % (args.batch, args.optimizer),
               "reference": spec["ref"], "value": sps, "unit": "samples/s", "ms_per_step": ms, "steps": args.steps,
               "warmup": did, "hip_graph": graphed, "final_loss": float(out[0].item()), "timing": spread(times, args.steps),
               "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tf / MFMA_PEAK_TFLOPS, "flop_per_sample": spec["flop_per_sample"],
                            "traffic": None,
                            # what the fp32 matrix pipe delivers on this chip under load (tools/micro/mfmabench.hip,
                            # profiles/r02_mfmabench.jsonl: one v_mfma_f32_32x32x2 per 64 cycles at the 2.13 GHz held)
                            "peak_measured": MFMA_MEASURED_TFLOPS, "frac_of_measured": tf / MFMA_MEASURED_TFLOPS}}
        del model
        torch.cuda.empty_cache()
        return res
    except Exception as exc:  # a failing extra leg must not take the headline down with it
        torch.cuda.synchronize()
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}


def fit_api(args, device, X, y, epochs=5, tile=10):
    """The public surface (reference basemodel.py:137-309): ``model.fit(x, y, batch_size=4096, epochs=E, verbose=0)`` on a
    device-resident dataset; samples/s over E epochs of a second call (the first call warms up: two eager steps, then the
    capture of the 16-step hipGraph fit() replays on groups of rows).  `value` is the DEFAULT call (shuffle=True) on the
    resident rows tiled `tile` times (640 steps per epoch): every epoch draws the reference's own permutation --
    ``torch.randperm(n, generator=...)`` on the HOST, exactly the RandomSampler of the reference's DataLoader
    (basemodel.py:213), so that both visit the rows in the same order; epoch e + 1's permutation is drawn while the GPU still
    runs epoch e.  ``shuffle_false`` is the same call without it; ``small_epochs`` the 4096 x 64 rows themselves (64-step
    epochs of ~6 ms of GPU time each, where the per-epoch host work -- the permutation above all -- is most of the wall time)."""
    import contextlib
    import io
    try:
        model = build_model(args, device)
        sink = io.StringIO()
        Xb, yb = X.repeat(tile, 1), y.repeat(tile)
        n = Xb.shape[0]
        steps = epochs * ((n - 1) // args.batch + 1)
        res = {"call": "model.fit(x, y, batch_size=%d, epochs=%d, verbose=0)" % (args.batch, epochs), "rows": n,
               "steps": steps, "unit": "samples/s", "steps_per_graph": int(os.environ.get("DCTR_FIT_STEPS_PER_GRAPH", "16"))}
        with contextlib.redirect_stdout(sink):
            model.fit(Xb, yb, batch_size=args.batch, epochs=1, verbose=0)
            for tag, shuffle in (("shuffle_true", True), ("shuffle_false", False)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                hist = model.fit(Xb, yb, batch_size=args.batch, epochs=epochs, verbose=0, shuffle=shuffle)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                res[tag] = {"value": epochs * n / dt, "ms_per_step": dt / steps * 1e3,
                            "last_epoch_loss": float(hist.history["loss"][-1])}
            n0, e0 = X.shape[0], 10
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.fit(X, y, batch_size=args.batch, epochs=e0, verbose=0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["small_epochs"] = {"rows": n0, "epochs": e0, "value": e0 * n0 / dt,
                                   "ms_per_step": dt / (e0 * ((n0 - 1) // args.batch + 1)) * 1e3}
        t0 = time.perf_counter()
        gen = torch.Generator().manual_seed(1)
        torch.randperm(n, generator=gen)
        res["host_randperm_ms_per_epoch"] = (time.perf_counter() - t0) * 1e3
        res["value"], res["ms_per_step"] = res["shuffle_true"]["value"], res["shuffle_true"]["ms_per_step"]
        del model, Xb, yb
        torch.cuda.empty_cache()
        return res
    except Exception as exc:
        torch.cuda.synchronize()
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}


def auto_steps_per_graph(steps):
    # (a replay boundary costs ~25 us, but 50 steps per graph measured 2.5 % SLOWER per step than 25: every captured step
    # has its own activation buffers, 40 MB each)
    divs = [d for d in range(8, 33) if steps % d == 0]
    return max(divs) if divs else min(16, steps)


def main():
    args = parse()
    if args.steps_per_graph <= 0:
        args.steps_per_graph = auto_steps_per_graph(args.steps)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (
            args.gpus, world, args.gpus))
    on_gpu = args.device == "cuda"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    device = ("cuda:%d" % local_rank) if on_gpu else "cpu"
    gpu_sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    dist = None
    if world > 1 or args.force_parallel:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    trace_buf = None
    if args.diag_trace:
        import ctypes
        from deepctr_torch._hip import lib as L
        L.use_diag_library()
        trace_buf = torch.zeros(3 * 4096 * 16, dtype=torch.int64, device=device)
        L.lib().dctr_dbg_mlp_trace(ctypes.c_void_p(trace_buf.data_ptr()))   # before any capture: graphs bake it in
    model = build_model(args, device)
    X, y = synth(args, device, rank)
    B = args.batch
    n_batches = X.shape[0] // B

    if dist is None:
        elapsed, out, graphed, did_warm, _, times = time_steps(model, X, y, B, args.steps, args.warmup,
                                                               args.steps_per_graph, not args.no_graph, args.repeats,
                                                               args.warmup_seconds)
        parallel = None
        S_blk = 0
    else:
        # table-sharded embeddings + data-parallel tower (SURVEY.md 8(e) option S); the compute between the
        # collectives is captured as hipGraph segments after a few eager steps
        from deepctr_torch import parallel as par
        # --exchange auto (the default) / try-direct: the direct exchange wherever it can be trusted -- one rank always; N > 1
        # after parallel.DirectExchange.self_test has passed on EVERY rank (a model-free check of the bytes kernels push into /
        # pull out of the peers' IPC-mapped buffers, three rounds over the same addresses) -- RCCL otherwise
        exchange, note = par.resolve_exchange(args.exchange, device) if on_gpu else ("rccl", "no GPU")
        exch_info = {"requested": args.exchange, "ran": exchange, "self_check": note}
        if rank == 0:     # which exchange the timed steps use, said BEFORE they run (stderr: stdout carries the JSON line)
            print("bench: sharded step over the %s exchange (requested %s; %s)" % (exchange, args.exchange, note),
                  file=sys.stderr, flush=True)
        parallel = par.ShardedTrainer(model, use_graphs=False, exchange=exchange)

        def batch(i):
            j = i % n_batches
            return X[j * B:(j + 1) * B], y[j * B:(j + 1) * B]

        def run(k):   # the trainer is told the next batch: its ids travel with this step's gradients
            return parallel.train_step(*batch(k), next_xb=batch(k + 1)[0])

        # direct exchange: S consecutive steps per hipGraph (ShardedTrainer.train_block), like the single-GPU runner; the
        # blocks are runs of S consecutive resident batches (a run that would wrap starts over at row 0)
        S_blk = min(args.steps_per_graph, args.steps, n_batches // 2) if (exchange == "direct" and on_gpu and
                                                                          not args.no_graph and
                                                                          os.environ.get("DCTR_SHARDED_BLOCK", "1") != "0") else 0
        if S_blk < 2 or args.steps % S_blk != 0:
            S_blk = 0
        blk_pos = [0]

        def block_at(j):
            return X[j * B:(j + S_blk) * B].view(S_blk, B, X.shape[1]), y[j * B:(j + S_blk) * B].view(S_blk, B)

        def run_block():
            j = blk_pos[0]
            jn = j + S_blk if j + 2 * S_blk <= n_batches else 0
            xs, ys = block_at(j)
            blk_pos[0] = jn
            return parallel.train_block(xs, ys, next_first=X[jn * B:(jn + 1) * B])

        def warm():
            nonlocal S_blk
            i = 0
            n_eager = min(3, args.warmup) if not args.no_graph else args.warmup
            for _ in range(n_eager):
                run(i)
                i += 1
            graphed = False
            if not args.no_graph and on_gpu:
                parallel.set_use_graphs(True)    # the compute segment is captured at the next step
                graphed = True
            if S_blk and graphed:
                for _ in range(3):      # eager block, capture, one replay
                    run_block()
                    i += S_blk
            else:
                S_blk = 0
                for _ in range(max(2 if graphed else 0, args.warmup - n_eager)):
                    run(i)
                    i += 1
            return i, graphed

        # The direct exchange has passed its byte-level self-test at this point, but at N > 1 the first real steps are the first
        # time the whole captured step crosses physical links: run the warm-up under a health check (a wait that timed out
        # raises a bit on the device; nothing in a direct-exchange step is a host collective, so every rank gets here) and let
        # ALL ranks fall back to RCCL together if any of them saw one.
        failed = 0
        try:
            i, graphed = warm()
            if exchange == "direct" and world > 1 and parallel._dx is not None:
                parallel._dx.check()
        except RuntimeError as exc:
            if not (exchange == "direct" and world > 1):
                raise
            failed = 1
            print("bench: rank %d: direct exchange failed in the warm-up steps (%s)" % (rank, exc), file=sys.stderr, flush=True)
        if exchange == "direct" and world > 1:
            flag = torch.tensor([failed], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                exchange = "rccl"
                exch_info["ran"], exch_info["fell_back"] = "rccl", "a direct-exchange wait timed out during the warm-up steps"
                if rank == 0:
                    print("bench: falling back to the RCCL exchange on every rank", file=sys.stderr, flush=True)
                try:
                    parallel.close()
                except Exception:
                    pass
                model = build_model(args, device)
                parallel = par.ShardedTrainer(model, use_graphs=False, exchange="rccl")
                S_blk = 0
                blk_pos[0] = 0
                i, graphed = warm()
        def block():       # exactly --steps steps between barrier + synchronize on both sides; MAX over ranks
            nonlocal i, out
            gpu_sync()
            dist.barrier()
            gpu_sync()
            t0 = time.perf_counter()
            if S_blk:
                for _ in range(args.steps // S_blk):
                    out = run_block()
                    i += S_blk
            else:
                for _ in range(args.steps):
                    out = run(i)
                    i += 1
            gpu_sync()
            dist.barrier()
            gpu_sync()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        out = None
        # time-based warm-up: the number of further blocks follows from the first one's MAX-reduced time, so every rank
        # runs the same number of collectives
        t_first = block()
        for _ in range(max(0, int(args.warmup_seconds / max(t_first, 1e-6)))):
            block()
        did_warm = i
        times = [block() for _ in range(max(1, args.repeats))]
        elapsed = sorted(times)[len(times) // 2]
    last_loss = float(out[0].item())
    model.model_plan().check_ids()
    if trace_buf is not None:
        import numpy as np
        np.save(args.diag_trace, trace_buf.view(3, 4096, 16).cpu().numpy())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        if not on_gpu:      # orchestration run: no kernels to time, no roofline
            print(json.dumps({"metric": "training samples/sec DeepFM Criteo batch=4096", "value": value, "unit": "samples/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
                              "higher_is_better": True, "scaling": "weak", "data": "synthetic", "device": "cpu (stand-ins)",
                              "timing": spread(times, args.steps), "final_loss": last_loss}), flush=True)
            if dist:
                dist.barrier()
                dist.destroy_process_group()
            return
        kern = time_hot_kernels(model, X, B, args.kernel_iters, args.optimizer)
        alg = algorithmic_bytes(B, args.optimizer)
        for k in kern:
            kern[k]["alg_bytes"] = alg[k]
            kern[k]["gbs"] = alg[k] / (kern[k]["avg_us"] * 1e-6) / 1e9
        # the HBM-bound kernels compete for "dominant"; the id-only segment pre-pass (latency / LDS-bound, run in the
        # tower's shadow) is listed with them in hot_path
        dom = max((k for k in kern if k != "embed_segments"), key=lambda k: kern[k]["avg_us"])
        # the step engine gathers inside the tower launch: the stand-alone gather kernel (still the lookup of every other
        # model and of predict()) is timed above but is not a launch of this step
        engine_on = parallel is None and model._fused_step_state() is not None and \
            model._fused_step_state().get("engine") is not None
        if engine_on:
            kern["embed_fwd"]["in_step"] = False
            dom = "embed_update"
        in_step = time_update_in_step(model, X, y, B) if (engine_on and dom == "embed_update") else None
        in_graph = time_kernels_in_graph(model, X, y, B) if engine_on else None
        clk = gpu_clock_power(lambda: time_steps(model, X, y, B, 200, 1, args.steps_per_graph, not args.no_graph)[0]) \
            if (world == 1 and on_gpu) else {}
        traffic = pmc_traffic(dom, args.optimizer, B)
        dom_us = in_step["avg_us"] if in_step else kern[dom]["avg_us"]
        dom_gbs = alg[dom] / (dom_us * 1e-6) / 1e9
        sat = saturating_launch(args, model, device) if (world == 1 and args.vocab >= 100000 and
                                                         not args.no_saturating) else None
        hot_us = sum(v["avg_us"] for v in kern.values())
        step_alg = sum(alg[k] for k in kern)
        # SURVEY.md 8(d)'s own per-sample figures: the whole train step (rows + ids + labels + dense parameters; Adagrad
        # adds the state's read + write) and the update proper (row + state, read + write) -- next to the per-kernel count
        # above, which also charges the re-read of the [B, ld] gradient strips, fm_s and the sorted keys
        rows_b = F_SPARSE * DIM * 4 + F_SPARSE * 4                                   # 1768 B: one deep + one wide row per field
        step_8d = (2 * rows_b + (F_SPARSE + N_DENSE) * 4 + 8 + 279) + (2 * rows_b if args.optimizer == "adagrad" else 0)
        upd_8d = (4 if args.optimizer == "adagrad" else 2) * rows_b
        # the tower launch (gather + tower forward + head + backward-data) is the step's LONGEST kernel and MFMA-bound:
        # forward + backward-data = 2 x 2 x (429 x 256 + 256 x 128 + 128) flop per sample
        k0 = F_SPARSE * DIM + N_DENSE
        tower_flop = 2 * 2 * (k0 * 256 + 256 * 128 + 128) * B
        tower = (in_step or {}).get("tower")
        dominant = None
        if tower:
            tf = tower_flop / (tower["avg_us"] * 1e-6) / 1e12
            dominant = {"kernel": "embed_tower_train", "bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "avg_us": tower["avg_us"],
                        "flop_per_launch": tower_flop, "timed": "in step", "detail": tower,
                        "note": "the step's longest launch (gather of the tile's rows + tower forward + head + BCE + "
                                "backward-data); the HBM-bound kernel the roofline object describes is the second longest"}
        result = {
            "metric": "training samples/sec DeepFM Criteo batch=4096", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DeepFM synthetic Criteo (26 sparse x %d vocab, 13 dense, emb_dim=16, batch=%d) "
                                   "fwd+bwd+%s, l2=0, dnn=(256,128)%s" % (args.vocab, B, args.optimizer,
                                                                         "" if args.ids == "uniform" else ", ids ~ Zipf(1.05) [secondary run]"),
                       "global_batch": world * B, "parallelism": ("tables sharded x%d + dp%d tower, %s exchange" % (world, world, parallel.exchange)) if parallel is not None else "single",
                       "hip_graph": bool(graphed), "steps_per_graph": (min(args.steps_per_graph, args.steps) if graphed and parallel is None else (S_blk or None) if graphed else None),
                       "eager_steps_in_timed_region": 0 if graphed else args.steps,
                       "warmup_steps_run": did_warm, "optimizer": args.optimizer,
                       "exchange": exch_info if parallel is not None else None},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dom_gbs / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("bytes"), "traffic_detail": traffic,
                         "alg_bytes_per_launch": alg[dom], "avg_us": dom_us,
                         "alg_bytes_per_sample": alg[dom] / B,
                         "alg_bytes_per_sample_8d": upd_8d if dom == "embed_update" else None,
                         "frac_8d": (upd_8d * B / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if dom == "embed_update" else None,
                         "dominant": dominant,
                         # scalar keys (a record that flattens nested objects keeps them): the step's LONGEST launch, timed
                         # inside hipGraph replays (dctr_stamp launches around it: time_kernels_in_graph), the update the
                         # same way, and the clock / power the box held while replaying train steps
                         "dominant_kernel": "embed_tower_train" if in_graph else None,
                         "dominant_bound": "mfma" if in_graph else None,
                         "dominant_avg_us": (in_graph or {}).get("tower_us"),
                         "dominant_frac": (tower_flop / (in_graph["tower_us"] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS)
                         if (in_graph and in_graph.get("tower_us")) else None,
                         "update_avg_us_in_graph": (in_graph or {}).get("update_us"),
                         "update_frac_in_graph": (alg["embed_update"] / (in_graph["update_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
                         if (in_graph and in_graph.get("update_us")) else None,
                         "in_graph": in_graph,
                         "sclk_mhz": clk.get("sclk_mhz"), "power_w": clk.get("power_w"),
                         "whole_step_frac_8d": value / world * step_8d / 1e9 / HBM_PEAK_GBS,
                         "whole_step_bytes_per_sample_8d": step_8d,
                         "timed": "in step" if in_step else "stand-alone", "in_step": in_step,
                         "standalone_avg_us": kern[dom]["avg_us"], "standalone_frac": kern[dom]["gbs"] / HBM_PEAK_GBS},
            "hot_path": {"kernels": kern, "sum_us": hot_us, "saturating": sat,
                         "frac_of_hbm_peak": step_alg / (hot_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "whole_step_frac_of_hbm_peak_kernel_bytes": step_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "whole_step_frac_8d": value / world * step_8d / 1e9 / HBM_PEAK_GBS},
            "final_loss": last_loss,
            "timing": dict(spread(times, args.steps), warmup_seconds=args.warmup_seconds,
                           protocol="value / ms_per_step = the median of `blocks` timed blocks of exactly `steps` steps"),
        }
        if world == 1 and not args.no_other_configs:
            del model
            torch.cuda.empty_cache()
            want = (lambda t: True) if args.legs == "all" else (lambda t: t in args.legs.split(","))
            result["other_configs"] = {name: other_config(name, args, device, X, y) for name in OTHER if want(name)}
            for tag in ("deepfm_varlen", "default_kwargs"):
                if not want(tag):
                    continue
                result["other_configs"][tag] = deepfm_leg(tag, args, device, X, y)
                if "ms_per_step" in result["other_configs"][tag]:
                    result["other_configs"][tag]["ms_per_step_vs_headline"] = result["other_configs"][tag]["ms_per_step"] / ms
            if want("fit_api"):
                result["other_configs"]["fit_api"] = fit_api(args, device, X, y)
            if parallel is None and want("sharded_1rank"):
                result["other_configs"]["sharded_1rank"] = sharded_1rank(args)
            if parallel is None and want("fit_api_sharded_1rank"):
                leg = result["other_configs"]["fit_api_sharded_1rank"] = fit_api_sharded_1rank(args)
                s1 = result["other_configs"].get("sharded_1rank", {})
                if "ms_per_step" in leg and "ms_per_step" in s1:
                    leg["ms_per_step_vs_sharded_1rank"] = leg["ms_per_step"] / s1["ms_per_step"]
            fa = result["other_configs"].get("fit_api", {})
            if "value" in fa:
                fa["vs_step_runner"] = fa["value"] / value
                fa["shuffle_false"]["vs_step_runner"] = fa["shuffle_false"]["value"] / value
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes "Librccl path : ..." through C stdio, which (piped)
        # would otherwise be flushed at exit, behind Python's line
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
