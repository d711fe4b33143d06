#!/usr/bin/env python
"""Where does model.fit() spend its host time at the bench shape?  cProfile over a second fit() call (the first one
warms up and captures), plus how many train steps ran eagerly vs inside replayed groups.
    python tools/fit_profile.py > gpurun_out/fit_profile.txt"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
dev = "cuda:0"
model = bench.build_model(args, dev)
X, y = bench.synth(args, dev, 0)
calls = {"eager": 0}
orig = model._train_step


def counted(xb, yb):
    if not torch.cuda.is_current_stream_capturing():
        calls["eager"] += 1
    return orig(xb, yb)


model._train_step = counted
sink = io.StringIO()
with contextlib.redirect_stdout(sink):
    model.fit(X, y, batch_size=args.batch, epochs=2, verbose=0)
torch.cuda.synchronize()
print("first fit(): eager train steps", calls["eager"])
calls["eager"] = 0
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(sink):
    pr.enable()
    model.fit(X, y, batch_size=args.batch, epochs=10, verbose=0)
    torch.cuda.synchronize()
    pr.disable()
dt = time.perf_counter() - t0
print("second fit(): %.1f ms for 640 steps = %.4f ms/step; eager train steps %d" % (dt * 1e3, dt / 640 * 1e3, calls["eager"]))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue()[:6000])

# one long epoch (10 x the rows): per-step rate of fit()'s group replays without the epoch boundaries
Xb, yb = X.repeat(10, 1), y.repeat(10)
for shuffle in (False, True):
    with contextlib.redirect_stdout(sink):
        model.fit(Xb, yb, batch_size=args.batch, epochs=1, verbose=0, shuffle=shuffle)       # (captures for this call shape)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit(Xb, yb, batch_size=args.batch, epochs=1, verbose=0, shuffle=shuffle)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("one epoch of 640 steps, shuffle=%s: %.1f ms = %.4f ms/step" % (shuffle, dt * 1e3, dt / 640 * 1e3))
# the same groups driven directly (no fit): 40 x step_rows + the loss accumulation fit() does
g = model._fit_graph["graph"]
tot = torch.zeros((), device=dev, dtype=torch.float64)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(40):
        outs = g.step_rows(Xb, yb, k * 16 * args.batch, None)
        tot += torch.stack([o[1].reshape(()) for o in outs]).double().sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("40 groups of 16 steps via step_rows: %.1f ms = %.4f ms/step" % (dt * 1e3, dt / 640 * 1e3))
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(40):
        outs = g.step_rows(Xb, yb, k * 16 * args.batch, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("40 groups, no loss accumulation: %.1f ms = %.4f ms/step" % (dt * 1e3, dt / 640 * 1e3))
