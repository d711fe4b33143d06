"""Whole-train-step HIP graphs.

At batch 4096 one DeepFM step is ~16 MB of compulsory HBM traffic (2 us at 8 TB/s) but 10-40 kernel
launches; eager PyTorch-ROCm pays 5-10 us of host time per launch.  Capturing forward + loss + backward
(with the fused sparse embedding update inside it) + the dense optimizer step into one hipGraph removes the
host from the loop: a step becomes (part of) one ``hipGraphLaunch``.  The kernels of
``libdctr_hip.so`` are enqueued on the caller's stream and never synchronise, so they are captured like
any ATen kernel.

Two measured facts shape this file (profiles/, MI355X):
  * a graph launch itself leaves the GPU idle for ~15-20 us between two replays -- 10 % of a 150 us step.  A graph
    therefore holds ``steps_per_graph`` consecutive train steps (each on its own static input buffers): the launch
    gap is paid once per group;
  * the batch reaches a graph through static buffers; there are TWO groups of buffers with one captured graph each
    (sharing one memory pool), and the copies of the next group's batches run on a side stream while the current
    graph executes, so they never sit on the critical path.
"""
import contextlib
import gc

import torch

from . import streams as _streams


@contextlib.contextmanager
def no_gc_during_capture():
    """A hipGraph capture must not be interrupted by Python's cyclic garbage collector: when it happens to free an
    older ``torch.cuda.CUDAGraph`` (a dead model's captured step sitting in a reference cycle) -- or anything else
    whose destructor calls into HIP -- inside the capture, HIP answers ``hipErrorStreamCaptureUnsupported`` from
    within a destructor and the process aborts (seen on MI355X / ROCm 7.2 as a rare crash of fit()).  Collect
    first, keep the collector off for the duration."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


class GraphedTrainStep(object):
    """Captures ``model._train_step`` for a fixed batch shape.

    ``step(xb, yb)`` (also ``__call__``) stages one batch; every ``steps_per_graph`` calls the staged group is
    launched as one graph.  The returned tensors are that step's static outputs: they hold the step's values once
    its group has been launched (``flush()`` launches / runs whatever is staged) and until the same slot is reused
    two groups later.
    """

    def __init__(self, model, x_example, y_example, steps_per_graph=1, double_buffer=True, inputs_ready=False):
        """``inputs_ready``: the caller guarantees that every batch it passes was COMPLETE on the device before the call
        (slices of a resident dataset).  Otherwise (default) the staging copies first wait for the caller's stream --
        a batch produced by an ``index_select`` that is still queued there must not be copied early."""
        self.model = model
        self.inputs_ready = bool(inputs_ready)
        self.S = max(1, int(steps_per_graph))
        self.n_slots = 2 if double_buffer else 1
        # one block per buffer group: the S batches of a group are rows of ONE tensor, so that a caller holding S
        # consecutive batches (a device-resident dataset) stages them with two copies instead of 2*S (step_block)
        self.xg = [torch.empty((self.S,) + tuple(x_example.shape), dtype=x_example.dtype, device=x_example.device)
                   for _ in range(self.n_slots)]
        self.yg = [torch.empty((self.S,) + tuple(y_example.shape), dtype=y_example.dtype, device=y_example.device)
                   for _ in range(self.n_slots)]
        self.x = [[self.xg[s][j] for j in range(self.S)] for s in range(self.n_slots)]
        self.y = [[self.yg[s][j] for j in range(self.S)] for s in range(self.n_slots)]
        self.graphs, self.outputs = [], []
        self.plan_version = None
        self._slot, self._j = 0, 0
        self._side = self._ready = self._free = self._free_ev = None

    @property
    def graph(self):
        return self.graphs[0] if self.graphs else None

    def capture(self, xb, yb):
        """Capture ``steps_per_graph`` steps per buffer group on (xb, yb).  Capture does not execute."""
        model = self.model
        plan = model.model_plan()
        plan.bind(xb.device)
        pool = None
        self.graphs, self.outputs = [], []
        for s in range(self.n_slots):
            for j in range(self.S):
                self.x[s][j].copy_(xb)
                self.y[s][j].copy_(yb)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            outs = []
            eng = (getattr(model, "_fused", None) or {}).get("engine")
            if eng is not None:
                eng._prepassed = None        # (a capture that was abandoned half-way must not lend its pre-pass to this one)
            with no_gc_during_capture(), torch.cuda.graph(g, pool=pool,
                                                          stream=_streams.side_stream(xb.device, "capture")):
                try:
                    for j in range(self.S):
                        model._defer_dense_join = j < self.S - 1      # (see BaseModel._train_step_fused)
                        # the batch of the NEXT captured step is known (it sits in the group's static block): the step
                        # engine enqueues its id-only pre-pass right behind this step's embedding update (_hip/step.py)
                        model._next_batch = self.x[s][j + 1] if j + 1 < self.S else None
                        outs.append(model._train_step(self.x[s][j], self.y[s][j]))
                finally:
                    model._defer_dense_join = False
                    model._next_batch = None
            if pool is None:
                pool = g.pool()
            self.graphs.append(g)
            self.outputs.append(outs)
        self.plan_version = plan.version
        self._side = _streams.side_stream(xb.device, "stage")
        self._ready = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._free_ev = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._free = [None] * self.n_slots
        self._slot, self._j = 0, 0
        return self

    def valid_for(self, xb):
        return (bool(self.graphs) and tuple(xb.shape) == tuple(self.x[0][0].shape) and
                self.plan_version == self.model.model_plan().version)

    def step(self, xb, yb):
        s, j = self._slot, self._j
        side = self._side
        if j == 0 and self._free[s] is not None:
            side.wait_event(self._free[s])          # the graph that last read this buffer group is done
        if not self.inputs_ready:
            side.wait_stream(torch.cuda.current_stream(xb.device))      # whatever produces xb / yb has run
        with torch.cuda.stream(side):
            self.x[s][j].copy_(xb, non_blocking=True)
            self.y[s][j].copy_(yb, non_blocking=True)
        if not self.inputs_ready:
            xb.record_stream(side)       # the caller may drop the batch right away: its memory must not be handed out
            yb.record_stream(side)       # again before the copy on the side stream has read it
        out = self.outputs[s][j]
        self._j += 1
        if self._j == self.S:
            self._launch()
        return out

    __call__ = step

    def step_block(self, x_block, y_block):
        """Stage and launch a whole group: ``x_block`` / ``y_block`` hold ``steps_per_graph`` consecutive batches
        (``[S*B, C]`` / ``[S*B]`` or any shape with the same element order).  Two copies per group instead of 2*S --
        at 8 steps per graph the 16 per-batch copies were 72 us between two graph launches.  Returns the outputs of
        the group's last step.  Only on a group boundary (nothing staged)."""
        if self._j != 0:
            raise RuntimeError("step_block() needs an empty group: flush() the staged steps first")
        s, side = self._slot, self._side
        if self._free[s] is not None:
            side.wait_event(self._free[s])
        if not self.inputs_ready:
            side.wait_stream(torch.cuda.current_stream(x_block.device))
        with torch.cuda.stream(side):
            self.xg[s].copy_(x_block.reshape(self.xg[s].shape), non_blocking=True)
            self.yg[s].copy_(y_block.reshape(self.yg[s].shape), non_blocking=True)
        if not self.inputs_ready:
            x_block.record_stream(side)
            y_block.record_stream(side)
        out = self.outputs[s][self.S - 1]
        self._j = self.S
        self._launch()
        return out

    def step_rows(self, X_all, y_all, lo, order=None):
        """Stage and launch a whole group straight from a device-resident dataset: rows ``[lo, lo + S*B)`` of
        ``X_all`` / ``y_all``, visited through ``order`` (a device permutation, ``fit(shuffle=True)``) when given -- ONE
        ``index_select`` per tensor and group into the group's static block, on the staging stream, instead of a pair
        per step on the training stream.  Returns the list of the group's S step outputs (static tensors: valid once the
        launch has run and until the same buffer group is reused two launches later)."""
        if self._j != 0:
            raise RuntimeError("step_rows() needs an empty group: flush() the staged steps first")
        s, side = self._slot, self._side
        n = self.S * self.x[s][0].shape[0]
        if self._free[s] is not None:
            side.wait_event(self._free[s])
        if not self.inputs_ready:
            side.wait_stream(torch.cuda.current_stream(X_all.device))
        with torch.cuda.stream(side):
            xg = self.xg[s].view((n,) + tuple(self.xg[s].shape[2:]))
            yg = self.yg[s].view((n,) + tuple(self.yg[s].shape[2:]))
            if order is None:
                xg.copy_(X_all[lo:lo + n], non_blocking=True)
                yg.copy_(y_all[lo:lo + n], non_blocking=True)
            else:
                idx = order[lo:lo + n]
                torch.index_select(X_all, 0, idx, out=xg)
                torch.index_select(y_all, 0, idx, out=yg)
        outs = self.outputs[s]
        self._j = self.S
        self._launch()
        return outs

    def sync_inputs(self):
        """With ``inputs_ready``: make the staging stream wait ONCE for what the caller's stream has enqueued so far (a
        new epoch's permutation), instead of before every group."""
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.x[0][0].device))

    def _launch(self):
        s = self._slot
        main = torch.cuda.current_stream(self.x[s][0].device)
        self._ready[s].record(self._side)
        main.wait_event(self._ready[s])
        self.graphs[s].replay()
        # a replay runs the captured dctr_lazy_apply launches without passing through LazyState.apply(): the host
        # flag that makes flush() / state_dict() / predict() bring every row up to date must be raised here too
        lazy = getattr(self.model.model_plan(), "_lazy", None)
        if lazy is not None:
            lazy.mark_dirty()
        self._free_ev[s].record(main)
        self._free[s] = self._free_ev[s]
        self._slot, self._j = (s + 1) % self.n_slots, 0

    def flush(self):
        """Run the steps staged so far (an incomplete group runs eagerly on the staged buffers).  Returns the last
        step's outputs, or None when nothing was staged."""
        s, j = self._slot, self._j
        if j == 0:
            return None
        main = torch.cuda.current_stream(self.x[s][0].device)
        self._ready[s].record(self._side)
        main.wait_event(self._ready[s])
        out = None
        for k in range(j):
            out = self.model._train_step(self.x[s][k], self.y[s][k])
        self._free_ev[s].record(main)
        self._free[s] = self._free_ev[s]
        self._slot, self._j = (s + 1) % self.n_slots, 0
        return out


def eager_warmup(model, batches):
    """Run the given (x, y) batches eagerly on a side stream, as torch's capture protocol wants."""
    side = _streams.side_stream(torch.device("cuda", torch.cuda.current_device()), "warm")
    side.wait_stream(torch.cuda.current_stream())
    outs = None
    with torch.cuda.stream(side):
        for xb, yb in batches:
            outs = model._train_step(xb, yb)
    torch.cuda.current_stream().wait_stream(side)
    return outs
