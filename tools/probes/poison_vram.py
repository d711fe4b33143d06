#!/usr/bin/env python
"""Fill most of the GPU's free memory with a NaN bit pattern and exit: processes started afterwards on the same box get
POISONED instead of zeroed memory from their first allocations on -- a fresh box hands out zeros, which hid this round's
out-of-bounds write of the CIN backward for a whole session.  Run it in front of the GPU suite."""
import torch

free, total = torch.cuda.mem_get_info()
n = int(free * 0.9) // 4
chunk = 1 << 30
left = n
bufs = []
while left > 0:
    k = min(chunk, left)
    t = torch.empty(k, dtype=torch.int32, device="cuda:0")
    t.fill_(0x7FC0DEAD)          # a quiet NaN as fp32, a large positive int32
    bufs.append(t)
    left -= k
torch.cuda.synchronize()
print("poisoned %.1f GB of %.1f GB" % (4.0 * n / 1e9, total / 1e9))
