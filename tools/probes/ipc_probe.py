"""Does torch's CUDA IPC work between processes that share ONE GPU here (HSA_ENABLE_IPC_MODE_LEGACY=0, dmabuf)?
N processes, gloo for the handshake; each maps every peer's buffer and writes its rank into its own slot of it."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.multiprocessing.reductions import reduce_tensor


def worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    buf = torch.zeros(world, 1024, device="cuda:0")
    words = torch.zeros(64, dtype=torch.int32, device="cuda:0")
    fn, args = reduce_tensor(buf)
    fn2, args2 = reduce_tensor(words)
    got = [None] * world
    dist.all_gather_object(got, (args, args2))
    peers = [fn(*g[0]) if r != rank else buf for r, g in enumerate(got)]
    pw = [fn2(*g[1]) if r != rank else words for r, g in enumerate(got)]
    for r in range(world):
        peers[r][rank].fill_(float(rank + 1))
        pw[r][rank:rank + 1].fill_(7)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    ok = all(float(buf[r].mean()) == r + 1 for r in range(world)) and words[:world].tolist() == [7] * world
    print("rank", rank, "ok" if ok else "MISMATCH", [float(buf[r, 0]) for r in range(world)], words[:world].tolist(), flush=True)
    # raw pointers work for kernels too?
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "deepctr-torch_amd"))
    import ctypes
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    step = torch.ones(1, dtype=torch.int32, device="cuda:0")
    ptrs = torch.tensor([p.data_ptr() + 4 * 32 for p in pw], dtype=torch.int64, device="cuda:0")
    err = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    s = L.stream_handle(torch.device("cuda:0"))
    L.check(lib.dctr_exchange_post(P(ptrs), world, rank, P(step), s))
    L.check(lib.dctr_exchange_wait(ctypes.c_void_p(words.data_ptr() + 4 * 32), world, P(step), 2000000, P(err), s))
    torch.cuda.synchronize()
    print("rank", rank, "post/wait err", int(err.item()), words[32:32 + world].tolist(), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, 29577), nprocs=world, join=True)
