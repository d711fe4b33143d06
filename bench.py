#!/usr/bin/env python
"""bench.py -- training samples/sec of the DeepCTR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward + BCE(sum) + backward (fused sparse embedding update) + dense optimizer step of DeepFM on
one batch of 4096 synthetic Criteo-shaped samples (26 sparse x 1M-row vocab, 13 dense, emb_dim 16) whose
dataset (4096 x 64 rows) is already resident in HBM.  Prints ONE JSON line (rank 0).  Besides the
contract's keys it carries
  roofline     achieved GB/s of the dominant hand-written kernel = ALGORITHMIC bytes per launch (DESIGN.md
               section 4) / its average duration measured here with HIP events on the launch stream
  cpu_baseline the reference's algorithm restated in torch-CPU (oracle/torch_port.py), timed on this box's
               host cores on a bounded sample (rank 0, N=1 only)
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
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replays")
    ap.add_argument("--steps-per-graph", type=int, default=8,
                    help="train steps captured per hipGraph (the ~15 us launch gap is paid once per graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-parallel", action="store_true", help="use the data-parallel trainer even with 1 rank")
    ap.add_argument("--kernel-iters", type=int, default=50, help="event-timed launches per hot-path kernel")
    ap.add_argument("--cpu-steps", type=int, default=3)
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
    ids = torch.randint(0, args.vocab, (n, F_SPARSE), generator=gen)           # uniform: worst case for caches
    X = torch.cat([ids.float(), torch.rand(n, N_DENSE, generator=gen)], dim=1).to(device)
    y = torch.randint(0, 2, (n,), generator=gen).float().to(device)
    return X, y


def algorithmic_bytes(B, opt):
    """Per-launch algorithmic HBM bytes of the hand-written kernels."""
    ld = (F_SPARSE * DIM + N_DENSE + 3) // 4 * 4
    x_row = (F_SPARSE + N_DENSE) * 4
    rows, wrows = F_SPARSE * DIM * 4, F_SPARSE * 4
    side = F_SPARSE * 4 + DIM * 4                                              # ids_t + fm_s side outputs
    fwd = B * (x_row + rows + wrows + ld * 4 + 8 + side)
    n_rw = 4 if opt == "adagrad" else 2                                        # table (+state): read + write
    # ids_t + g_out + fm_s + g_fm + g_wide, then the row read-modify-writes (FM's backward is folded algebraically:
    # the forward's copy of the rows is not re-read)
    upd = B * (F_SPARSE * 4 + rows + DIM * 4 + 8 + n_rw * (rows + wrows))
    return {"embed_fwd": fwd, "embed_update": upd}


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
              torch.empty(len(plan.units), B, dtype=torch.int32, device=dev), torch.empty(B, DIM, device=dev))
             for j in range(ring)]

    def fwd(j):
        Xb, out, ids_t, fm_s = slots[j % ring]
        L.check(lib.dctr_embed_fwd(cplan, _ptr(Xb), Xb.stride(0), B, _ptr(out), plan.ld_out, _ptr(wide), 1, _ptr(fm),
                                   None, plan.units_ptr(), len(plan.units), _ptr(ids_t), _ptr(fm_s), DIM, s))

    ws, ws_n = plan.update_workspace(B, dev)

    def upd(j):
        Xb, out, ids_t, fm_s = slots[j % ring]
        L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), len(plan.units), plan.max_vocab, _ptr(ids_t), B,
                                      _ptr(g_out), plan.ld_out, _ptr(out), plan.ld_out, _ptr(fm_s), DIM, _ptr(g_fm),
                                      _ptr(g_wide), 1, L.UPD_ADAGRAD if opt == "adagrad" else L.UPD_SGD, lr, eps,
                                      None, 0, None, _ptr(ws), ws_n, s))

    stages = [("embed_fwd", fwd), ("embed_update", upd)]
    for j in range(ring):       # every slot's side outputs exist before any update is timed
        fwd(j)
    for j in range(ring):
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


def pmc_traffic(kernel, opt, B):
    """HBM bytes per launch of a hand-written kernel from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs, KiB units, corrected with factors calibrated on launches of
    known byte counts as MI355X_MICROARCH.md prescribes for access patterns other than wide streaming reads).
    Returns None when no PMC summary for this kernel / launch size is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        d = json.load(fh)
    tag = "embed_fwd" if kernel == "embed_fwd" else "embed_update_%s" % opt
    # the PMC driver launches every kernel at B = 4096 and B = 32768: the smaller grid is this bench's launch
    cands = sorted((int(k.split("@grid")[1]), v) for k, v in d.get("kernels", {}).items() if k.split("@grid")[0] == tag)
    if not cands or B != 4096:
        return None
    best = cands[0][1]
    return {"bytes": best["fetch_bytes_gather_corrected"] + best["write_bytes_corrected"],
            "fetch_raw": best["fetch_raw_bytes"], "fetch_gather_calibrated": best["fetch_bytes_gather_corrected"],
            "fetch_x2_streaming_rule": best["fetch_bytes_x2"], "write": best["write_bytes_corrected"],
            "source": "profiles/r01_pmc_traffic.json"}


def cpu_baseline(args):
    """The reference's dense-gradient algorithm (torch-CPU port) on this box's host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from torch_port import DeepFMPort, make_optimizer, train_step
    torch.manual_seed(0)
    port = DeepFMPort(F_SPARSE, args.vocab, DIM, N_DENSE, hidden=(256, 128))
    opt = make_optimizer(port, args.optimizer)
    gen = torch.Generator().manual_seed(0)
    X = torch.cat([torch.randint(0, args.vocab, (args.batch, F_SPARSE), generator=gen).float(),
                   torch.rand(args.batch, N_DENSE, generator=gen)], dim=1)
    y = torch.randint(0, 2, (args.batch,), generator=gen).float()
    train_step(port, opt, X, y)  # warm-up
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        train_step(port, opt, X, y)
    dt = time.perf_counter() - t0
    return {"value": args.batch * args.cpu_steps / dt, "unit": "samples/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d train steps (after 1 warm-up) of the same DeepFM/batch=%d/vocab=%d workload, dense [V,D] "
                      "gradients + dense torch.optim.%s like the reference, l2=0; host has %d logical cpus" % (
                          args.cpu_steps, args.batch, args.vocab, args.optimizer, os.cpu_count() or 0),
            "ms_per_step": dt / args.cpu_steps * 1e3}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (
            args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    dist = None
    if world > 1 or args.force_parallel:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    model = build_model(args, device)
    X, y = synth(args, device, rank)
    B = args.batch
    n_batches = X.shape[0] // B
    parallel = None
    if dist is not None:
        # table-sharded embeddings + data-parallel tower (SURVEY.md 8(e) option S): 3 all-to-alls + 1 all-reduce per
        # step; the compute between the collectives is captured as hipGraph segments after a few eager steps
        from deepctr_torch import parallel as par
        parallel = par.ShardedTrainer(model, use_graphs=False)

    def batch(i):
        j = i % n_batches
        return X[j * B:(j + 1) * B], y[j * B:(j + 1) * B]

    step_fn = (lambda xb, yb: parallel.train_step(xb, yb)) if parallel else (lambda xb, yb: model._train_step(xb, yb))

    def run(k):
        """Step on batch k.  The sharded trainer is told the next batch: its ids travel with this step's gradients."""
        if parallel is not None:
            return parallel.train_step(*batch(k), next_xb=batch(k + 1)[0])
        return step_fn(*batch(k))

    use_graph = (not args.no_graph) and parallel is None
    n_eager = min(3, args.warmup) if not args.no_graph else args.warmup
    i = 0
    for _ in range(n_eager):
        run(i)
        i += 1
    graphed = None
    if use_graph:
        from deepctr_torch._hip.graph import GraphedTrainStep
        try:
            # (the synthetic dataset was uploaded and synchronised long before: its slices are complete)
            graphed = GraphedTrainStep(model, *batch(0), steps_per_graph=args.steps_per_graph,
                                       inputs_ready=True).capture(*batch(i))
            step_fn = graphed
        except Exception as exc:  # capture is an optimisation: report and continue eagerly
            print("hipGraph capture failed (%s: %s); running eager" % (type(exc).__name__, exc), file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()
    if parallel is not None and not args.no_graph:
        parallel.use_graphs = True       # segments re-capture on the next step
        parallel._shape = None
        graphed = "segments"
    for _ in range(max(0, args.warmup - n_eager)):
        run(i)
        i += 1

    if use_graph and graphed is not None and graphed != "segments":
        graphed.flush()                  # the timed region starts on a group boundary
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block = 0
    if use_graph and graphed is not None and graphed != "segments":
        block = graphed.S if (n_batches % graphed.S == 0 and graphed.S > 1) else 0
    done = 0
    while done < args.steps:
        j = i % n_batches
        if block and args.steps - done >= block and j + block <= n_batches and graphed._j == 0:
            # the dataset is resident and the group's batches are consecutive rows: stage them with two copies
            out = graphed.step_block(X[j * B:(j + block) * B], y[j * B:(j + block) * B])
            i += block
            done += block
        else:
            out = run(i)
            i += 1
            done += 1
    if use_graph and graphed is not None and graphed != "segments":
        tail = graphed.flush()           # K % steps_per_graph leftover steps run eagerly, inside the timed region
        out = tail if tail is not None else out
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last_loss = float(out[0].item())
    model.model_plan().check_ids()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        kern = time_hot_kernels(model, X, B, args.kernel_iters, args.optimizer)
        alg = algorithmic_bytes(B, args.optimizer)
        for k in kern:
            kern[k]["alg_bytes"] = alg[k]
            kern[k]["gbs"] = alg[k] / (kern[k]["avg_us"] * 1e-6) / 1e9
        dom = max(kern, key=lambda k: kern[k]["avg_us"])
        traffic = pmc_traffic(dom, args.optimizer, B)
        hot_us = sum(v["avg_us"] for v in kern.values())
        step_alg = sum(alg[k] for k in kern)
        result = {
            "metric": "training samples/sec DeepFM Criteo batch=4096", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DeepFM synthetic Criteo (26 sparse x %d vocab, 13 dense, emb_dim=16, batch=%d) "
                                   "fwd+bwd+%s, l2=0, dnn=(256,128)" % (args.vocab, B, args.optimizer),
                       "global_batch": world * B, "parallelism": ("tables sharded x%d + dp%d tower" % (world, world)) if parallel is not None else "single",
                       "hip_graph": graphed is not None, "steps_per_graph": (args.steps_per_graph if use_graph and graphed is not None else None),
                       "optimizer": args.optimizer},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["gbs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": kern[dom]["gbs"] / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("bytes"), "traffic_detail": traffic,
                         "alg_bytes_per_launch": alg[dom], "avg_us": kern[dom]["avg_us"]},
            "hot_path": {"kernels": kern, "sum_us": hot_us,
                         "frac_of_hbm_peak": step_alg / (hot_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "whole_step_frac_of_hbm_peak": step_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "final_loss": last_loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
