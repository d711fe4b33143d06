#!/usr/bin/env python
"""On-box measurements that size the design: HBM stream bandwidth, random 64-B / 4-B
row gathers, fp32 atomic scatter, launch / graph-replay overhead, hipBLASLt fp32 GEMMs at the MLP shapes, and
the hand-written embed kernels at batch 4096 and at a saturating batch.  Writes one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepctr-torch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

dev = "cuda:0"
res = {}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return {"avg_us": sum(ts) / len(ts), "min_us": ts[0], "med_us": ts[len(ts) // 2]}


props = torch.cuda.get_device_properties(0)
res["device"] = {"name": props.name, "cus": props.multi_processor_count, "mem_gb": props.total_memory / 2**30,
                 "n_devices": torch.cuda.device_count(), "host_cpus": os.cpu_count()}

# 1. stream copy / read
n = 1 << 30
a = torch.empty(n // 4, device=dev, dtype=torch.float32).normal_()
b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a), 10)
res["copy_4GiB_rw"] = dict(t, gbs=2 * n / (t["min_us"] * 1e-6) / 1e9)
t = timeit(lambda: a.sum(), 10)
res["read_sum_1GiB"] = dict(t, gbs=n / (t["min_us"] * 1e-6) / 1e9)
del b

# 2. random row gathers over a 1.6 GB table
V, D = 26_000_000, 16
table = torch.empty(V, D, device=dev).normal_()
for rows in (106_496, 27_262_976 // 8):
    idx = torch.randint(0, V, (rows,), device=dev)
    t = timeit(lambda: table.index_select(0, idx), 10)
    res["torch_gather_64B_rows_%d" % rows] = dict(t, gbs=rows * 64 * 2 / (t["min_us"] * 1e-6) / 1e9)
w1 = torch.empty(V, 1, device=dev).normal_()
idx = torch.randint(0, V, (106_496,), device=dev)
t = timeit(lambda: w1.index_select(0, idx), 10)
res["torch_gather_4B_rows_106496"] = t
src = torch.randn(106_496, D, device=dev)
t = timeit(lambda: table.index_add_(0, idx, src), 10)
res["torch_index_add_64B_rows_106496"] = t
del table, w1

# 3. launch overheads
x = torch.zeros(64, device=dev)
t0 = time.perf_counter()
for _ in range(2000):
    x.add_(1)
torch.cuda.synchronize()
res["eager_tiny_kernel_us_per_launch"] = (time.perf_counter() - t0) / 2000 * 1e6
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    for _ in range(40):
        x.add_(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
res["graph_replay_40_tiny_kernels_us"] = (time.perf_counter() - t0) / 200 * 1e6

# 4. fp32 GEMMs of the MLP / CIN shapes through torch (hipBLASLt / rocBLAS)
for (M, K, N) in [(4096, 429, 256), (4096, 256, 128), (4096, 10413, 128), (65536, 1664, 128), (65536, 676, 128),
                  (4096, 4096, 4096)]:
    A, Bm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    t = timeit(lambda: A @ Bm, 10)
    res["gemm_f32_%dx%dx%d" % (M, K, N)] = dict(t, tflops=2 * M * K * N / (t["min_us"] * 1e-6) / 1e12)

# 5. hand-written embed kernels at B=4096 and at a saturating batch
import bench  # noqa: E402


class A:
    vocab, batch, optimizer = 1_000_000, 4096, "adagrad"


for opt in ("adagrad", "sgd"):
    A.optimizer = opt
    model = bench.build_model(A, dev)
    for Bsz in (4096, 32_768, 262_144):     # 262 144 = the saturating launch of SURVEY 8(d)
        A.batch = Bsz
        gen = torch.Generator().manual_seed(0)
        X = torch.cat([torch.randint(0, A.vocab, ((8 if Bsz <= 32_768 else 2) * Bsz, 26), generator=gen).float(), torch.rand((8 if Bsz <= 32_768 else 2) * Bsz, 13, generator=gen)],
                      1).to(dev)
        k = bench.time_hot_kernels(model, X, Bsz, 20 if Bsz <= 32_768 else 6, opt, ring=8 if Bsz <= 32_768 else 2)
        alg = bench.algorithmic_bytes(Bsz, opt)
        for name in k:
            k[name]["gbs"] = alg[name] / (k[name]["min_us"] * 1e-6) / 1e9
            k[name]["alg_bytes"] = alg[name]
        res["embed_kernels_%s_B%d" % (opt, Bsz)] = k
    del model
    torch.cuda.empty_cache()

print(json.dumps(res, indent=1))
