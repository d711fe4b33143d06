"""CPU, gloo: ``bench.py --gpus 2`` end to end -- argument handling, rendezvous from RANK / WORLD_SIZE / MASTER_*, the
table-sharded trainer with announced batches, time-based warm-up with the same number of collectives on every rank, the
timed blocks between barriers, MAX over ranks, ONE JSON line from rank 0 -- with the device kernels stood in for
(tests/mock_lib.py, tests/shard_standin.py), so that the first run on an 8-GPU node is not the first run of this code.
(bench.py itself has no CPU compute path: ``--device cpu`` only swaps RCCL for gloo and drops the torch.cuda calls.)"""
import io
import json
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "deepctr-torch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    torch.set_num_threads(1)
    from test_sharded_gloo import _patch_for_cpu
    _patch_for_cpu()
    from deepctr_torch import parallel as par
    from shard_standin import TorchShardOps
    par.HipShardOps = TorchShardOps            # the four device steps of the exchange, in torch
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--device", "cpu", "--steps", "3", "--warmup", "2", "--batch", "32",
                "--vocab", "60", "--repeats", "2", "--warmup-seconds", "0.05", "--no-cpu-baseline", "--no-other-configs"]
    buf = io.StringIO()
    old = sys.stdout
    sys.stdout = buf
    try:
        bench.main()
    finally:
        sys.stdout = old
    with open(os.path.join(out_dir, "rank%d.out" % rank), "w") as fh:
        fh.write(buf.getvalue())


def test_bench_runs_two_ranks_end_to_end_over_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out0 = open(os.path.join(str(tmp_path), "rank0.out")).read().strip().splitlines()
    out1 = open(os.path.join(str(tmp_path), "rank1.out")).read().strip()
    assert out1 == "", "only rank 0 prints"
    lines = [ln for ln in out0 if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["timing"]["blocks"] == 2 and d["ms_per_step"] > 0
    assert abs(d["value"] - 2 * 32 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]      # whole-job samples / s
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 100                   # finite BCE(sum) of 32 samples
