#!/usr/bin/env python
"""Event-timed tower kernels of the fused train step (csrc/mlp.hip) at one shape, stand-alone, through the C ABI:
    k_mlp_train (forward + head + BCE + backward-data)  |  k_mlp_wgrad + k_mlp_reduce (+ in-kernel Adagrad)
with a torch fp64 check of every output, a sweep over the batch slices of the weight-gradient kernel (diag library
only: DCTR_WGRAD_SLICES) and -- with --trace -- the per-workgroup phase stamps of the diag build.

    python tools/tower_bench.py [--shape deepfm|xdeepfm] [--batch 4096] [--iters 40] [--slices 5,6,7,8,10,14] [--trace]
Prints one JSON object."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch  # noqa: E402

from deepctr_torch._hip import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="deepfm")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--slices", default="")
ap.add_argument("--trace", action="store_true")
ap.add_argument("--diag", action="store_true")
ap.add_argument("--flush-mb", type=int, default=0, help="--trace: stream this many MB through the chip before the traced launch")
ap.add_argument("--nx", type=int, default=8, help="input buffers the launches rotate over")
args = ap.parse_args()
if args.trace or args.slices or args.diag:
    L.use_diag_library()
lib = L.lib()
dev = "cuda:0"
SHAPES = {"deepfm": (429, (256, 128)), "xdeepfm": (429, (256, 256)), "dcn": (429, (128, 128)), "pnn": (754, (128, 128))}
K, hidden = SHAPES[args.shape]
B = args.batch


def r4(n):
    return (n + 3) // 4 * 4


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


torch.manual_seed(0)
ldx = r4(K)
NX = args.nx                                          # rotate the inputs: nothing is served from a warm L2 by accident
xs = [torch.randn(B, ldx, device=dev) for _ in range(NX)]
ys = [torch.randint(0, 2, (B,), device=dev).float() for _ in range(NX)]
part0 = torch.randn(B, device=dev) * 0.1
part1 = torch.randn(B, device=dev) * 0.1
bias = torch.randn(1, device=dev) * 0.1
# one flat slab for parameters / gradients / Adagrad state, rows padded to 4 floats (what dense.DenseSlab builds)
dims, Kl = [], K
for N in hidden:
    dims.append((N, Kl, r4(Kl)))
    Kl = N
off, layout = 0, []
for (N, Kin, ld) in dims:
    layout.append(("W", off, N, Kin, ld))
    off += N * ld
    layout.append(("b", off, N, 0, 0))
    off += r4(N)
layout.append(("wo", off, hidden[-1], 0, 0))
off += r4(hidden[-1])
layout.append(("bias", off, 1, 0, 0))
off += 4
slab_p = torch.zeros(off, device=dev)
slab_g = torch.zeros(off, device=dev)
slab_s = torch.zeros(off, device=dev)
Ws, bs = [], []
for kind, o, N, Kin, ld in layout:
    if kind == "W":
        w = slab_p[o:o + N * ld].view(N, ld)
        w[:, :Kin].normal_(0, 0.05)
        Ws.append((w, o, N, Kin, ld))
    elif kind == "b":
        slab_p[o:o + N].normal_(0, 0.05)
        bs.append((slab_p[o:o + N], o))
    elif kind == "wo":
        slab_p[o:o + N].normal_(0, 0.05)
        wo, wo_off = slab_p[o:o + N], o
    else:
        slab_p[o:o + 1].copy_(bias)
        bias_p, bias_off = slab_p[o:o + 1], o
p0 = slab_p.clone()
hs = [torch.empty(B, r4(N), device=dev) for N in hidden]
dhs = [torch.empty_like(h) for h in hs]
y_pred = torch.empty(B, device=dev)
loss = torch.empty((), device=dev)
g_logit = torch.empty(B, device=dev)
gx = torch.empty(B, ldx, device=dev)


def make_desc():
    d = L.Mlp()
    d.n_layers = len(hidden)
    for l, (w, o, N, Kin, ld) in enumerate(Ws):
        e = d.layer[l]
        e.W, e.bias, e.h, e.dh = w.data_ptr(), bs[l][0].data_ptr(), hs[l].data_ptr(), dhs[l].data_ptr()
        e.gW, e.gbias = slab_g[o:].data_ptr(), slab_g[bs[l][1]:].data_ptr()
        e.K, e.N, e.ld_w, e.ld_h, e.relu = Kin, N, ld, hs[l].stride(0), 1
    d.w_out, d.g_w_out = wo.data_ptr(), slab_g[wo_off:].data_ptr()
    return d


desc = make_desc()
step = L.DenseStep()
step.kind, step.lr, step.eps = L.UPD_ADAGRAD, 0.01, 1e-10
step.grad_base, step.param_base, step.state_base = slab_g.data_ptr(), slab_p.data_ptr(), slab_s.data_ptr()
s = L.stream_handle(torch.device(dev))
res = {"shape": args.shape, "K": K, "hidden": list(hidden), "batch": B}


def launch_train(i, ws):
    L.check(lib.dctr_mlp_train_step(ctypes.byref(desc), ptr(xs[i % NX]), ldx, B, ptr(part0), ptr(part1), ptr(bias_p),
                                    ptr(ys[i % NX]), ptr(y_pred), ptr(loss), ptr(g_logit), ptr(slab_g[bias_off:]),
                                    ptr(gx), ldx, ptr(ws), 1, None, s), "train_step")


def launch_wgrad(i, ws, with_step):
    L.check(lib.dctr_mlp_train_wgrad(ctypes.byref(desc), ptr(xs[i % NX]), ldx, B, ptr(g_logit), ptr(ws), ptr(loss),
                                     ptr(slab_g[bias_off:]), ctypes.byref(step) if with_step else None, s), "wgrad")


def timed(fn, iters):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters):
        ev[i][0].record()
        fn(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return {"min_us": round(ts[0], 2), "median_us": round(ts[len(ts) // 2], 2), "avg_us": round(sum(ts) / len(ts), 2)}


def check(ws):
    """one train + wgrad (no optimizer step) against torch fp64"""
    slab_p.copy_(p0)
    launch_train(0, ws)
    launch_wgrad(0, ws, False)
    torch.cuda.synchronize()
    x = xs[0][:, :K].double().requires_grad_(True)
    h = x
    Wd = [w[:, :Kin].double().requires_grad_(True) for (w, o, N, Kin, ld) in Ws]
    bd = [b[0].double().requires_grad_(True) for b in bs]
    wod = wo.double().requires_grad_(True)
    biasd = bias_p.double().requires_grad_(True)
    for W, b in zip(Wd, bd):
        h = torch.relu(h @ W.t() + b)
    z = part0.double() + part1.double() + h @ wod + biasd
    lossd = torch.nn.functional.binary_cross_entropy(torch.sigmoid(z), ys[0].double(), reduction="sum")
    lossd.backward()
    errs = {"loss_rel": abs(float(loss) - float(lossd)) / abs(float(lossd)),
            "y_pred": float((y_pred.double() - torch.sigmoid(z)).abs().max()),
            "gx": float((gx[:, :K].double() - x.grad).abs().max() / x.grad.abs().max())}
    for l, (w, o, N, Kin, ld) in enumerate(Ws):
        gW = slab_g[o:o + N * ld].view(N, ld)
        errs["gW%d" % l] = float((gW[:, :Kin].double() - Wd[l].grad).abs().max() / Wd[l].grad.abs().max())
        errs["gW%d_pad" % l] = float(gW[:, Kin:].abs().max()) if ld > Kin else 0.0
        gb = slab_g[bs[l][1]:bs[l][1] + N]
        errs["gb%d" % l] = float((gb.double() - bd[l].grad).abs().max() / bd[l].grad.abs().max())
    errs["gwo"] = float((slab_g[wo_off:wo_off + hidden[-1]].double() - wod.grad).abs().max() / wod.grad.abs().max())
    errs["gbias"] = float((slab_g[bias_off].double() - biasd.grad[0]).abs() / biasd.grad.abs().max())
    return errs


def workspace():
    n = lib.dctr_mlp_train_workspace_floats(ctypes.byref(desc), B)
    return torch.empty(max(1, n), device=dev), n


ws, n_ws = workspace()
res["check"] = check(ws)
res["worst_rel_err"] = max(v for k, v in res["check"].items() if not k.endswith("_pad"))
for i in range(5):
    launch_train(i, ws)
    launch_wgrad(i, ws, True)
torch.cuda.synchronize()
res["train"] = timed(lambda i: launch_train(i, ws), args.iters)
res["wgrad_reduce"] = timed(lambda i: launch_wgrad(i, ws, True), args.iters)
res["train+wgrad_reduce"] = timed(lambda i: (launch_train(i, ws), launch_wgrad(i, ws, True)), args.iters)
res["workspace_floats"] = int(n_ws)
flop_train = 2.0 * B * sum(N * Kin for (w, o, N, Kin, ld) in Ws) * 2 + 2.0 * B * hidden[-1] * 2
flop_wgrad = 2.0 * B * sum(N * Kin for (w, o, N, Kin, ld) in Ws)
res["train_tflops"] = round(flop_train / (res["train"]["median_us"] * 1e-6) / 1e12, 1)
res["wgrad_tflops"] = round(flop_wgrad / (res["wgrad_reduce"]["median_us"] * 1e-6) / 1e12, 1)

if args.slices:
    sweep = {}
    for S in [int(v) for v in args.slices.split(",")]:
        os.environ["DCTR_WGRAD_SLICES"] = str(S)
        ws2, n2 = workspace()
        e = check(ws2)
        for i in range(3):
            launch_wgrad(i, ws2, True)
        torch.cuda.synchronize()
        t = timed(lambda i: launch_wgrad(i, ws2, True), args.iters)
        t["worst_rel_err"] = max(v for k, v in e.items() if not k.endswith("_pad"))
        t["workspace_floats"] = int(n2)
        sweep[S] = t
    os.environ.pop("DCTR_WGRAD_SLICES")
    res["wgrad_slices_sweep"] = sweep

if args.trace:
    buf = torch.zeros(3 * 4096 * 16, dtype=torch.int64, device=dev)
    lib.dctr_dbg_mlp_trace(ctypes.c_void_p(buf.data_ptr()))
    if args.flush_mb:
        junk = torch.empty(args.flush_mb << 18, device=dev)
        junk2 = torch.empty_like(junk)
        torch.cuda.synchronize()
        junk2.copy_(junk)
    launch_train(0, ws)
    launch_wgrad(0, ws, True)
    torch.cuda.synchronize()
    lib.dctr_dbg_mlp_trace(None)
    t = buf.view(3, 4096, 16).cpu().numpy().astype(np.int64)

    def st(v):
        v = np.asarray(v, dtype=np.float64) * 0.01
        return {"mean": round(float(v.mean()), 2), "p50": round(float(np.percentile(v, 50)), 2),
                "p90": round(float(np.percentile(v, 90)), 2), "max": round(float(v.max()), 2)}

    f = t[0][t[0][:, 0] > 0]          # forward part of k_mlp_train (two-layer towers: slots 0..8, 15)
    b = t[1][t[1][:, 0] > 0]          # its backward part
    tr = {"n_wg": int(len(f))}
    if len(f) and len(b) == len(f):
        t0 = f[:, 0].min()
        tr["start_spread"] = st(f[:, 0] - t0)
        tr["fwd_stage_x"] = st(f[:, 1] - f[:, 0])
        tr["fwd_layer0_mfma"] = st(f[:, 2] - f[:, 1])
        tr["fwd_layer0_epilogue"] = st(f[:, 3] - f[:, 2])
        tr["fwd_layer0_barrier"] = st(f[:, 4] - f[:, 3])
        tr["fwd_layer1_mfma"] = st(f[:, 5] - f[:, 4])
        tr["fwd_layer1_epilogue"] = st(f[:, 6] - f[:, 5])
        tr["fwd_layer1_barrier"] = st(f[:, 7] - f[:, 6])
        tr["fwd_projection"] = st(f[:, 15] - f[:, 7])
        tr["fwd_total"] = st(f[:, 15] - f[:, 0])
        tr["head"] = st(b[:, 0] - f[:, 15])
        tr["bwd_stage_top"] = st(b[:, 1] - b[:, 0])
        tr["bwd_layer1_mfma+epi"] = st(b[:, 2] - b[:, 1])
        tr["bwd_layer1_barrier"] = st(b[:, 3] - b[:, 2])
        tr["bwd_layer0_mfma+epi"] = st(b[:, 4] - b[:, 3])
        tr["bwd_layer0_barrier"] = st(b[:, 5] - b[:, 4])
        tr["bwd_total"] = st(b[:, 15] - b[:, 0])
        tr["wg_total"] = st(b[:, 15] - f[:, 0])
        tr["end"] = st(b[:, 15] - t0)
    res["trace_train"] = tr
    w = t[2][t[2][:, 0] > 0]
    if len(w):
        t0 = w[:, 0].min()
        wk = w[w[:, 1] > 0]
        res["trace_wgrad"] = {"n_wg": int(len(w)), "n_gemm_wg": int(len(wk)), "start": st(wk[:, 0] - t0),
                              "mainloop": st(wk[:, 1] - wk[:, 0]), "lds_park+barrier": st(wk[:, 2] - wk[:, 1]),
                              "combine+store": st(wk[:, 3] - wk[:, 2]), "wg_total": st(wk[:, 3] - wk[:, 0]),
                              "end": st(wk[:, 3] - t0)}
print(json.dumps(res, indent=1))
