#!/usr/bin/env python
"""Where fit() through the sharded trainer at one rank spends its time: a fixed cost per call or a cost per epoch?
    DCTR_FIT_FORCE_TRAINER=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29411 python tools/probes/fit_sharded_overhead.py"""
import os, sys, time, io, contextlib, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.argv = ["bench.py"]
import torch
import bench as b
args = b.parse()
torch.cuda.set_device(0)
model = b.build_model(args, "cuda:0")
X, y = b.synth(args, "cuda:0", 0)
X, y = X.repeat(4, 1), y.repeat(4)
n = X.shape[0]
def run(epochs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X, y, batch_size=args.batch, epochs=epochs, verbose=0, shuffle=False)
    torch.cuda.synchronize(); return time.perf_counter() - t0
print("first call (1 epoch)  %.1f ms" % (run(1) * 1e3))
for e in (1, 1, 3, 6):
    dt = run(e)
    print("epochs %d: %.1f ms total, %.3f ms/step" % (e, dt * 1e3, dt / (e * (n // args.batch)) * 1e3))
pr = cProfile.Profile(); pr.enable(); run(3); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
