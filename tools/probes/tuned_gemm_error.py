#!/usr/bin/env python
"""Rounding error of the library's default pick and of TunableOp's pick for the three GEMMs of FiBiNET's 10 413-wide first
layer, against float64 (rms and max of the absolute error; same operands)."""
import json
import os
import sys
import time

import torch

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
B, K, N = 4096, 10413, 128
x = torch.randn(B, K, device=dev, generator=g) * 0.3
W = torch.randn(N, K, device=dev, generator=g) * 0.02
b = torch.randn(N, device=dev, generator=g) * 0.1
gh = torch.randn(B, N, device=dev, generator=g) * 0.01


def ops():
    return {"fwd addmm [B,K]x[K,N]": lambda: torch.addmm(b, x, W.t()),
            "bwd-data mm [B,N]x[N,K]": lambda: torch.mm(gh, W),
            "wgrad mm [N,B]x[B,K]": lambda: torch.mm(gh.t(), x)}


ref = {"fwd addmm [B,K]x[K,N]": torch.addmm(b.double(), x.double(), W.double().t()),
       "bwd-data mm [B,N]x[N,K]": torch.mm(gh.double(), W.double()),
       "wgrad mm [N,B]x[B,K]": torch.mm(gh.double().t(), x.double())}


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 20 * 1e6


out = {}
for mode in ("default", "tuned"):
    t = torch.cuda.tunable
    t.enable(mode == "tuned")
    t.tuning_enable(mode == "tuned")
    for name, fn in ops().items():
        y = fn()
        e = (y.double() - ref[name]).abs()
        out.setdefault(name, {})[mode] = {"rms_err": float(e.pow(2).mean().sqrt()), "max_err": float(e.max()),
                                          "ref_rms": float(ref[name].pow(2).mean().sqrt()), "us": round(timed(fn), 1)}
print(json.dumps(out, indent=1))
