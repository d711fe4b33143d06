"""Whole-train-step HIP graph.

At batch 4096 one DeepFM step is ~16 MB of compulsory HBM traffic (2 us at 8 TB/s) but ~40 kernel
launches; eager PyTorch-ROCm pays 5-10 us of host time per launch.  Capturing forward + loss + backward
(with the fused sparse embedding update inside it) + the dense optimizer step into one hipGraph removes the
host from the loop: a step becomes one ``hipGraphLaunch``.  The kernels of
``libdctr_hip.so`` are enqueued on the caller's stream and never synchronise, so they are captured like
any ATen kernel.
"""
import torch


class GraphedTrainStep(object):
    """Captures ``model._train_step`` for a fixed batch shape.

    ``warm`` batches are first run eagerly on a side stream (real training steps on real data -- nothing is
    replayed twice), then one more step is captured.  ``__call__`` copies a batch into the static buffers and
    replays.  Returned tensors are static: read them before the next call.
    """

    def __init__(self, model, x_example, y_example):
        self.model = model
        self.x = torch.empty_like(x_example)
        self.y = torch.empty_like(y_example)
        self.graph = None
        self.outputs = None
        self.plan_version = None

    def capture(self, xb, yb):
        """Capture one step on (xb, yb).  Capture does not execute: call ``replay`` afterwards."""
        model = self.model
        self.x.copy_(xb)
        self.y.copy_(yb)
        plan = model.model_plan()
        plan.bind(self.x.device)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = model._train_step(self.x, self.y)
        self.plan_version = plan.version
        return self

    def valid_for(self, xb):
        return (self.graph is not None and tuple(xb.shape) == tuple(self.x.shape) and
                self.plan_version == self.model.model_plan().version)

    def __call__(self, xb, yb):
        self.x.copy_(xb)
        self.y.copy_(yb)
        self.graph.replay()
        return self.outputs


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
