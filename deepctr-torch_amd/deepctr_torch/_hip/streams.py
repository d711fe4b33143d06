"""The side streams of this package, created ONCE per device and all at the same time.

``torch.cuda.Stream()`` hands out the 32 streams of a per-device pool round-robin: the 33rd object is the first stream
again.  Streams created per model (the update's pre-pass stream, the weight-gradient fork, the batch-staging stream of a
graphed step, torch's own default capture stream) therefore start to ALIAS each other once a process has built a dozen
models -- two "different" streams of one hipGraph capture are then one queue: a stream waits for itself, a copy meant
for outside the capture lands inside it, and the first replay takes the process down (found as a segmentation fault
that needed 14 earlier tests in the same process).  Every role below gets its own pool stream, created back to back at
the first request -- distinct by construction -- and captures run on the package's own capture stream instead of
torch's default one.
"""
import torch

ROLES = ("capture", "seg", "fork", "stage", "warm", "shard", "sweep")
_STREAMS = {}


def side_stream(device, role):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("side streams exist on the GPU only")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    per = _STREAMS.get(idx)
    if per is None:
        per = _STREAMS[idx] = {r: torch.cuda.Stream(device=torch.device("cuda", idx)) for r in ROLES}
        ids = set(s.stream_id for s in per.values())
        if len(ids) != len(ROLES):          # (another library exhausted the pool in between: cannot happen back to back)
            raise RuntimeError("torch handed out aliasing side streams")
    return per[role]
