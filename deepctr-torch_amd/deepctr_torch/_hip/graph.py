"""Whole-train-step HIP graph.

At batch 4096 one DeepFM step is ~16 MB of compulsory HBM traffic (2 us at 8 TB/s) but 10-40 kernel
launches; eager PyTorch-ROCm pays 5-10 us of host time per launch.  Capturing forward + loss + backward
(with the fused sparse embedding update inside it) + the dense optimizer step into one hipGraph removes the
host from the loop: a step becomes one ``hipGraphLaunch``.  The kernels of
``libdctr_hip.so`` are enqueued on the caller's stream and never synchronise, so they are captured like
any ATen kernel.

The batch reaches the graph through static input buffers.  There are TWO buffer sets with one captured graph
each (sharing one memory pool): the copy of batch k+1 into set B runs on a side stream while the graph of batch k
still reads set A, so the two device-to-device copies (~10 us, 6 % of a DeepFM step) leave the critical path.
"""
import torch


class GraphedTrainStep(object):
    """Captures ``model._train_step`` for a fixed batch shape.

    ``__call__`` copies a batch into the next static buffer set (side stream) and replays that set's graph.
    Returned tensors are static: read them before the next call.
    """

    def __init__(self, model, x_example, y_example, double_buffer=True):
        self.model = model
        self.n_slots = 2 if double_buffer else 1
        self.x = [torch.empty_like(x_example) for _ in range(self.n_slots)]
        self.y = [torch.empty_like(y_example) for _ in range(self.n_slots)]
        self.graphs, self.outputs = [], []
        self.plan_version = None
        self._i = 0
        self._side = None
        self._ready = None
        self._free = None

    # kept for callers that look at the first slot
    @property
    def graph(self):
        return self.graphs[0] if self.graphs else None

    def capture(self, xb, yb):
        """Capture one step per buffer set on (xb, yb).  Capture does not execute: call the object afterwards."""
        model = self.model
        plan = model.model_plan()
        plan.bind(xb.device)
        pool = None
        self.graphs, self.outputs = [], []
        for s in range(self.n_slots):
            self.x[s].copy_(xb)
            self.y[s].copy_(yb)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                out = model._train_step(self.x[s], self.y[s])
            if pool is None:
                pool = g.pool()
            self.graphs.append(g)
            self.outputs.append(out)
        self.plan_version = plan.version
        self._side = torch.cuda.Stream(device=xb.device)
        self._ready = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._free = [None] * self.n_slots
        self._free_ev = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._i = 0
        return self

    def valid_for(self, xb):
        return (bool(self.graphs) and tuple(xb.shape) == tuple(self.x[0].shape) and
                self.plan_version == self.model.model_plan().version)

    def __call__(self, xb, yb):
        s = self._i % self.n_slots
        self._i += 1
        main = torch.cuda.current_stream(xb.device)
        side = self._side
        if self._free[s] is not None:
            side.wait_event(self._free[s])          # the graph that last read this buffer set is done
        with torch.cuda.stream(side):
            self.x[s].copy_(xb, non_blocking=True)
            self.y[s].copy_(yb, non_blocking=True)
            self._ready[s].record(side)
        main.wait_event(self._ready[s])
        self.graphs[s].replay()
        self._free_ev[s].record(main)
        self._free[s] = self._free_ev[s]
        return self.outputs[s]


def eager_warmup(model, batches):
    """Run the given (x, y) batches eagerly on a side stream, as torch's capture protocol wants."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    outs = None
    with torch.cuda.stream(side):
        for xb, yb in batches:
            outs = model._train_step(xb, yb)
    torch.cuda.current_stream().wait_stream(side)
    return outs
