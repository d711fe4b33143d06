#!/usr/bin/env python
"""dctr_bilinear_wide_bwd alone at the Criteo shape (26 fields x 16, 128 hidden units, batch 4096): time per call
(pack + main + reduce) over hipEvents, after 40 warm-up calls (the first variant timed was 10 % slow otherwise).
With the diagnostics library (make -C deepctr-torch_amd/csrc diag) DCTR_WIDE_VAR="<2|1><VAR>" selects a timing variant of
the main kernel ("1...": the field count read at run time instead of the F = 26 instantiation).  The kernel's VAR bits:
1 no barriers, 2 no partial stores, 4 no ring re-loads, 8 no LDS gradient writes, 16 phase stamps (s_memtime sums per wave:
MFMA block / barrier / behind it, and prologue / loop / epilogue), 32 ring re-loads out of L1, 64 schedule entries carried
in registers, 128 part 2 at raised priority -- results are wrong for every bit but 16, the time shows what each costs.  The
host entry instantiates 0, 15 and 16 (compile time); the single-bit variants of profiles/r06_bilinear_wide_variants.txt were
measured with their instantiations added to the #ifdef DCTR_DIAG block of dctr_bilinear_wide_bwd.
    python tools/probes/wide_bwd_probe.py [variants...]      e.g.  20 215 216 10"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
import torch
from deepctr_torch._hip import lib as L
from deepctr_torch._hip import ops

variants = sys.argv[1:] or ["20"]
if variants != ["20"] or os.environ.get("DCTR_PROBE_DIAG"):
    L.use_diag_library()
lib = L.lib()
B, F, D, H = int(os.environ.get("B", 4096)), 26, 16, 128
P = F * (F - 1) // 2
dev = "cuda:0"
torch.manual_seed(0)
meta = ops.BilinearMeta(F, "interaction")
sched4, pair_w = meta.wide_tables(dev)
E = torch.randn(B, F, D, device=dev); V = torch.randn(B, F, D, device=dev)
Wf = torch.randn(P, D, D, device=dev) * 0.3
gh = torch.randn(B, H, device=dev); W0 = torch.randn(H, 2 * P * D + 13, device=dev) * 0.05
gE = torch.empty(B, F, D, device=dev); gV = torch.empty_like(gE); gW = torch.empty(P, D, D, device=dev)
ws = torch.empty(lib.dctr_bilinear_wide_bwd_workspace_floats(B, P), device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def call():
    L.check(lib.dctr_bilinear_wide_bwd(p(E), F * D, p(V), F * D, p(Wf), p(sched4), sched4.shape[0], p(pair_w), P, P, F, D, B,
                                       p(gh), H, p(W0), W0.stride(0), H, p(gE), p(gV), p(gW), p(ws), None,
                                       L.stream_handle(torch.device(dev))), "wide")
flop = 2.0 * B * 2 * P * (16 * H + 3 * 256)
for _ in range(40):           # (clocks settle: the first variant timed was 10 % slow otherwise)
    call()
torch.cuda.synchronize()
for v in variants:
    os.environ["DCTR_WIDE_VAR"] = v
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    n = 30
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / n
    if v.endswith("16") and len(v) == 3:      # phase stamps: per-wave cycle sums in the partials' first floats
        off = lib.dctr_bilinear_wide_bwd_workspace_floats(B, P) - ((B + 15) // 16) * P * 256
        for t in (0, 100, 255):
            d = ws[off + t * P * 256: off + t * P * 256 + 64].reshape(8, 8).cpu()
            ng = float(d[0, 3])
            print("  tile %3d  cycles per group (MFMA block / barrier / behind it) per wave: " % t +
                  "  ".join("%d/%d/%d" % (d[w, 0] / ng, d[w, 1] / ng, d[w, 2] / ng) for w in range(8)))
            print("            whole launch (prologue / loop / behind the loop): " +
                  "  ".join("%d/%d/%d" % (d[w, 4], d[w, 5], d[w, 6]) for w in (0, 4)))
    print("variant %-4s  %8.1f us per call (pack + main + reduce)   %.1f TFLOP/s on the MFMA work" % (v, us, flop / us / 1e6))
