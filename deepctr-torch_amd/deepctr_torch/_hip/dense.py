"""One flat slab for every dense (non-table) parameter, its gradient and its optimizer state.

The reference steps ``torch.optim`` over the parameter list: for Adagrad that is addcmul / sqrt / add /
addcdiv ``foreach`` launches plus a ``zero_grad`` per step (basemodel.py:244,262) -- ~8 launches of pure latency
for 143 k dense parameters.  Here the parameters are re-seated ONCE as views of one contiguous fp32 slab (values
preserved, ``state_dict`` keys and shapes unchanged), the gradient kernels write into a second slab with the same
layout, and ``dctr_dense_opt`` (csrc/head.hip) applies SGD / Adagrad to the whole slab in one launch.

Weights consumed by the MFMA tower (csrc/mlp.hip) get their rows padded to a multiple of 4 floats inside the
slab (the parameter becomes a strided view ``slab[N, ld][:, :K]``); the padding stays zero because its
gradient is written as zero.
"""
import ctypes

import torch

from . import lib as L
from . import streams as _streams


def _r4(n):
    return (int(n) + 3) // 4 * 4


class DenseSlab(object):
    L2_VALUE_SMALL = 1 << 16        # regularised elements up to which reg_value() is the single-workgroup launch

    def __init__(self, params, pad_rows=()):
        """params: list of nn.Parameter (fp32, same device); pad_rows: parameters whose rows are padded."""
        self.params = list(params)
        pad = set(id(p) for p in pad_rows)
        dev = self.params[0].device
        self._lay = {}
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("the dense slab holds float32 parameters of one device")
            if id(p) in pad and p.dim() == 2 and p.shape[1] > 1:
                rows, cols, ld = p.shape[0], p.shape[1], _r4(p.shape[1])
            else:
                rows, cols, ld = 1, p.numel(), p.numel()
            self._lay[id(p)] = (off, rows, cols, ld)
            off += _r4(rows * ld)
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.state = None
        self.state2 = None       # Adam: exp_avg_sq (state = exp_avg)
        self.lam = None          # per-element lambda of the L2 terms on slab parameters (set_l2), or None
        self.steps = None        # device int32: optimizer steps completed (Adam's bias correction; graph-replay safe)
        self._optimizer = None
        with torch.no_grad():
            for p in self.params:
                v = self._view(self.flat, p)
                v.copy_(p.detach())
                p.data = v
                p.grad = None
        self._ptrs = [p.data_ptr() for p in self.params]
        # fork / join of the weight-gradient kernels (mlp.TowerHeadFunction): see fork_stream()
        self.overlap = False
        self._fork = None
        self._pending = None
        self.deferred = None      # overlap == "defer": the closure that enqueues the weight-gradient kernels
        self.after_update = None  # topology "tower_side": enqueues the forked weight gradients behind the update's launch
        # in-kernel optimizer step (single-GPU fused train step): the kernels that finish the dense gradients (the
        # tower's weight-gradient reduction, the update kernel's Linear.weight workgroups) also step the parameters,
        # and the embedding update runs on a side stream beside them -- see begin_inline_step()
        self.inline = None        # DenseStep while a fused train step with in-kernel optimizer is being assembled
        self.inline_done = False  # the kernels of this step applied it: step() has nothing left to do
        self.wgrad_side = False   # topology of the in-kernel-optimizer step: True = weight gradients on the fork stream
        self.wgrad_on_seg = False  # ... and that fork stream is the pre-pass's ("tower_seg")
        self.flag_sync = False    # topology "flags": gather_side with the two cross-queue edges replaced by dctr_step_wait
        self.sync_timeout_us = 20000
        self._sync = None         # the sync block (int32[16], zero at rest between steps' signal / wait pairs)
        self.upd_keep = None      # tower_seg: the operands of the last embedding update (alive until the next one is enqueued)
        self.gather_side = False  # ... True = gather AND update on the pre-pass's stream (ops.EmbedFunction.forward)
        self.main_keep = None     # gather_side: tensors the main stream's weight-gradient kernels of the last step read
        self.update_stream = None  # the side stream of this step's segment pre-pass (set by ops.EmbedFunction.forward):
        #                            the tower + head launch makes it wait for itself, the update then runs there

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_lay"] = [self._lay[id(p)] for p in self.params]     # id() keys do not survive pickling
        d["_fork"] = d["_pending"] = d["deferred"] = d["after_update"] = None   # streams / events / closures: per process
        d["overlap"] = False
        d["inline"] = d["update_stream"] = d["main_keep"] = d["upd_keep"] = d["_sync"] = None
        d["flag_sync"] = False
        d["wgrad_side"] = d["gather_side"] = d["wgrad_on_seg"] = False
        d["_fork_events"] = None
        d["inline_done"] = False
        return d

    # ---- fork / join ------------------------------------------------------------------------------------------------
    # The dense gradients are consumed by nothing but step(); the kernels that produce them may therefore run on a
    # second stream beside the embedding update (two latency-bound kernels of ~30 us each at batch 4096).  The
    # producer asks for the stream with fork_stream() (None unless the train step switched `overlap` on), reports what
    # it enqueued with forked(), and join() -- called by step() and by whoever else reads the gradient slab -- makes
    # the current stream wait for it.  Inside a hipGraph capture the fork and the join become graph edges.
    # ``overlap == "defer"`` (parallel.ShardedTrainer): the producer enqueues nothing and leaves a closure in
    # ``deferred``; the trainer calls it with a stream of its choice AFTER its graph segment, so that the weight
    # gradients overlap the gradient all-to-all and the embedding update, and joins before the dense all-reduce.  The
    # closure addresses the tensors of the run that created it -- under a hipGraph those are the capture's static
    # buffers, so the closure of the capture stays valid for every replay.
    def begin_inline_step(self, kind, lr, eps):
        """The coming train step applies plain SGD / Adagrad to the slab INSIDE the gradient kernels (no
        ``dctr_dense_opt`` launch, no cross-queue join in front of it).  Returns False (nothing armed) when the slab
        carries L2 terms or Adam state -- those keep the separate regularised optimizer kernel."""
        if kind not in ("sgd", "adagrad") or self.lam is not None:
            return False
        if kind == "adagrad" and self.state is None:
            return False
        st = L.DenseStep()
        st.kind = L.UPD_ADAGRAD if kind == "adagrad" else L.UPD_SGD
        st.lr, st.eps = float(lr), float(eps)
        st.grad_base = self.grad.data_ptr()
        st.param_base = self.flat.data_ptr()
        st.state_base = self.state.data_ptr() if kind == "adagrad" else None
        self.inline, self.inline_done, self.update_stream = st, False, None
        return True

    def end_inline_step(self):
        self.inline = None
        self.update_stream = None

    def fork_stream(self, device, force=False):
        if not force and (not self.overlap or self.overlap == "defer" or torch.device(device).type != "cuda"):
            return None
        if self._pending is not None:
            self.join()
        if self._fork is None:
            self._fork = _streams.side_stream(device, "fork")
        return self._fork

    def side_chain_open(self, side):
        """True when ``side`` still carries the previous step's un-joined update and is part of the running hipGraph
        capture: work enqueued on it now is ordered behind that update without waiting for the current stream."""
        p = self._pending
        if p is None or side is None or p[0] is not side or not torch.cuda.is_current_stream_capturing():
            return False
        with torch.cuda.stream(side):
            return bool(torch.cuda.is_current_stream_capturing())

    def sync_block(self, device):
        """The step's device-side dependency words (include/dctr.h: dctr_step_wait).  Allocated once, outside any
        hipGraph capture's pool."""
        if self._sync is None or self._sync.device != torch.device(device):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the step's sync block must exist before a hipGraph capture begins (run one eager step)")
            self._sync = torch.zeros(L.SYNC_INTS, dtype=torch.int32, device=device)
        return self._sync

    def check_sync(self, reset=False):
        """Raise if a dctr_step_wait ever timed out (synchronises the device).  ``reset``: zero the block -- after an
        exception interrupted a step between a signal and its wait."""
        if self._sync is None:
            return
        torch.cuda.synchronize(self._sync.device)
        bad = int(self._sync[L.SYNC_ERR].item())
        if reset or bad:
            self._sync.zero_()
            torch.cuda.synchronize(self._sync.device)
        if bad:
            raise RuntimeError("a device-side step dependency timed out (signals %s): kernels of the two queues were not "
                               "running concurrently (a profiler serialising kernels?) -- results since then are invalid; "
                               "set DCTR_STEP_TOPOLOGY=update_side" % bin(bad))

    def fork_event(self, k):
        """Two events that live as long as the slab (an event created inside a hipGraph capture and collected during a
        later one aborts the process: graph.no_gc_during_capture)."""
        ev = getattr(self, "_fork_events", None)
        if ev is None:
            ev = self._fork_events = [torch.cuda.Event(), torch.cuda.Event()]
        return ev[k]

    def forked(self, stream, keep_alive, done=None):
        """``done``: an event recorded behind the forked work -- join() then waits for IT instead of the stream's tail
        (the stream goes on to carry work the joiner must not wait for)."""
        self._pending = (stream, keep_alive, done)

    def join(self):
        p = self._pending
        self.main_keep = None
        if p is not None:
            self._pending = None
            cur = torch.cuda.current_stream(self.flat.device)
            if len(p) > 2 and p[2] is not None:
                cur.wait_event(p[2])
            else:
                cur.wait_stream(p[0])
        # (the tensors of keep_alive are released only now, on the stream that has just waited for their last reader)

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._lay = {id(p): lay for p, lay in zip(self.params, d["_lay"])}

    def _rows(self, buf, p):
        off, rows, cols, ld = self._lay[id(p)]
        return buf[off:off + rows * ld].view(rows, ld)

    def _view(self, buf, p):
        off, rows, cols, ld = self._lay[id(p)]
        if rows > 1 or ld != cols:
            return buf[off:off + rows * ld].view(rows, ld)[:, :cols]
        return buf[off:off + cols].view(p.shape)

    def intact(self):
        """False once somebody re-allocated a parameter (``model.to()``, ``p.data = ...``): re-adopt then."""
        return all(p.data_ptr() == q for p, q in zip(self.params, self._ptrs))

    def grad_of(self, p):
        """Where the gradient kernels write: the padded ``[rows, ld]`` block for padded weights, else a view
        shaped like the parameter."""
        if id(p) not in self._lay:
            return None
        off, rows, cols, ld = self._lay[id(p)]
        if rows > 1 or ld != cols:
            return self._rows(self.grad, p)
        return self.grad[off:off + cols].view(p.shape)

    def attach_grads(self):
        """``param.grad`` = its view of the gradient slab (what a user or a generic optimizer would read)."""
        for p in self.params:
            p.grad = self._view(self.grad, p)

    def adopt_adagrad_state(self, optimizer):
        """Move ``optimizer.state[p]['sum']`` of every slab parameter into a third slab (values kept), so that
        ``optimizer.state_dict()`` keeps working and the fused kernel updates the very same memory."""
        self.state = torch.zeros_like(self.flat)
        for p in self.params:
            st = optimizer.state.get(p)
            if st is None or "sum" not in st:
                raise RuntimeError("Adagrad state missing for a dense parameter")
            v = self._view(self.state, p)
            v.copy_(st["sum"])
            st["sum"] = v
        # the padding of the state stays 0; its gradient is 0, so sqrt(0) + eps never divides anything but 0

    def adopt_adam_state(self, optimizer):
        """``exp_avg`` / ``exp_avg_sq`` of every slab parameter become views of two state slabs (torch creates Adam's
        state at its first step(): it is created here, zeros, when missing); ``step`` is kept in ``self.steps`` on the
        device and written back to the optimizer's state by ``sync_optimizer_state()``."""
        self.state = torch.zeros_like(self.flat)
        self.state2 = torch.zeros_like(self.flat)
        t0 = 0
        for p in self.params:
            st = optimizer.state[p]
            for key, slab in (("exp_avg", self.state), ("exp_avg_sq", self.state2)):
                v = self._view(slab, p)
                if key in st:
                    v.copy_(st[key])
                st[key] = v
            if "step" in st:
                t0 = max(t0, int(float(st["step"])))
            else:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
        self.steps = torch.full((1,), t0, dtype=torch.int32, device=self.flat.device)
        self._optimizer = optimizer

    def sync_optimizer_state(self):
        """Adam's per-parameter ``step`` entries follow the device counter (call outside graph capture)."""
        if self._optimizer is None or self.steps is None:
            return
        t = float(int(self.steps.item()))
        for p in self.params:
            st = self._optimizer.state.get(p)
            if st is not None and "step" in st:
                st["step"] = torch.tensor(t, dtype=torch.float32)

    def set_l2(self, lam_of):
        """``{param: lambda}`` of the L2 terms on slab parameters: applied inside the optimizer kernel as
        g += 2*lambda*p (what autograd adds for ``lambda * sum(p^2)``, basemodel.py:412-428)."""
        lam_of = dict((id(p), float(v)) for p, v in lam_of.items() if v > 0)
        self._l2_items = None
        if not lam_of:
            self.lam = None
            return
        self.lam = torch.zeros_like(self.flat)
        todo = []
        for p in self.params:
            if id(p) in lam_of:
                self._view(self.lam, p).fill_(lam_of[id(p)])
                todo.append((self._view(self.flat, p), lam_of[id(p)]))
        # few regularised elements (DeepFM's defaults: the 13 dense weights of Linear): their logged value is ONE
        # single-workgroup launch over those tensors instead of a product + a two-kernel dot over the whole slab
        if self.flat.is_cuda and sum(v.numel() for v, _ in todo) <= self.L2_VALUE_SMALL and \
                all(v.is_contiguous() for v, _ in todo):
            items = (L.DenseItem * len(todo))()
            for i, (v, lam) in enumerate(todo):
                items[i].p, items[i].g, items[i].state, items[i].n, items[i].l2 = v.data_ptr(), None, None, v.numel(), lam
            self._l2_items = items
            self._l2_out = torch.zeros((1,), dtype=torch.float32, device=self.flat.device)

    def reg_value(self):
        """sum(lambda * p^2) over the slab (the dense parameters' share of get_regularization_loss), or None."""
        if self.lam is None:
            return None
        if getattr(self, "_l2_items", None) is not None:
            out = torch.empty_like(self._l2_out)
            L.check(L.lib().dctr_l2_value_multi(self._l2_items, len(self._l2_items), ctypes.c_void_p(out.data_ptr()),
                                                L.stream_handle(self.flat.device)), "dctr_l2_value_multi")
            return out
        return torch.dot(self.lam * self.flat, self.flat).reshape(1)

    def _opt_reg(self, kind, lr, eps, beta1, beta2, stream):
        o = L.LazyOpt()
        o.kind = {"sgd": L.LAZY_SGD, "adagrad": L.LAZY_ADAGRAD, "adam": L.LAZY_ADAM}[kind]
        o.lr, o.eps, o.beta1, o.beta2 = float(lr), float(eps), float(beta1), float(beta2)
        if self.steps is None:
            self.steps = torch.zeros((1,), dtype=torch.int32, device=self.flat.device)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        L.check(L.lib().dctr_dense_opt_reg(ptr(self.flat), ptr(self.grad), ptr(self.state), ptr(self.state2),
                                           ptr(self.lam), self.numel, ctypes.byref(o), ptr(self.steps), stream),
                "dctr_dense_opt_reg")
        L.check(L.lib().dctr_lazy_step_inc(ptr(self.steps), stream), "dctr_lazy_step_inc")

    def step(self, kind, lr, eps=0.0, beta1=0.0, beta2=0.0):
        if self.inline_done:          # the gradient kernels of this step already applied it (begin_inline_step);
            self.inline_done = False  # whoever reads the parameters next joins the fork (join())
            return
        self.join()
        if kind == "adam" or self.lam is not None:
            # Adam, or L2 terms on slab parameters: the regularised kernel of csrc/lazy.hip (one launch + a counter)
            if kind == "adagrad" and self.state is None:
                raise RuntimeError("adopt_adagrad_state() first")
            if kind == "adam" and self.state2 is None:
                raise RuntimeError("adopt_adam_state() first")
            self._opt_reg(kind, lr, eps, beta1, beta2, L.stream_handle(self.flat.device))
            return
        opt = L.UPD_ADAGRAD if kind == "adagrad" else L.UPD_SGD
        st = self.state
        if opt == L.UPD_ADAGRAD and st is None:
            raise RuntimeError("adopt_adagrad_state() first")
        L.check(L.lib().dctr_dense_opt(ctypes.c_void_p(self.flat.data_ptr()), ctypes.c_void_p(self.grad.data_ptr()),
                                       ctypes.c_void_p(st.data_ptr()) if st is not None else None, self.numel, opt,
                                       float(lr), float(eps), L.stream_handle(self.flat.device)), "dctr_dense_opt")
