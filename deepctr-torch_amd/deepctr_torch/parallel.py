# -*- coding: utf-8 -*-
"""Data-parallel training: one process per GPU, RCCL over xGMI (``torch.distributed`` backend ``"nccl"``).

The reference's only multi-GPU path is ``torch.nn.DataParallel`` inside ONE process
(``basemodel.py:206-209``): every step it re-broadcasts all parameters (1.77 GB at the Criteo shape), scatters
the batch, gathers outputs and reduce-adds dense ``[V, D]`` gradients onto ``gpus[0]``.  Here each rank keeps a
replica and trains on its own shard of the global batch; per step there are exactly two exchanges:

  dense parameters (0.57 MB for DeepFM)   one flat bucket, ``all_reduce(SUM)`` -- SUM, not mean, because the
                                          reference's loss is ``reduction='sum'`` over the global batch
                                          (``basemodel.py:209,254``);
  embedding rows                          each rank folds FM's backward into per-sample row gradients and
                                          all-gathers one packed payload ``[B, G | g_wide | ids]``; every
                                          replica then runs the SAME deterministic fused update
                                          (``dctr_embed_update``, csrc/update.hip) over the global batch.
                                          The kernel has no atomics and sums duplicate ids in (id, sample)
                                          order, so replicas stay bit-identical without any parameter
                                          broadcast -- which is why float atomics were not an option here.

Semantics = the reference's: one optimizer step on the gradient summed over ``world_size x batch`` samples.

The exchange logic (payload layout, bucket views, collectives) is plain torch and is exercised on CPU with the
``gloo`` backend in ``tests/test_parallel_gloo.py``; only the kernels need a GPU.
"""
import torch
import torch.distributed as dist


class DenseBucket(object):
    """All dense (non-table) gradients live in ONE flat fp32 buffer; ``param.grad`` are views into it, so the
    all-reduce needs no packing copies and is a single collective (the whole DeepFM tower is 0.57 MB:
    latency-bound, so one message beats per-tensor messages)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def attach(self):
        """Zero the bucket and point every ``param.grad`` at its view (call between forward and backward)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, group=None, async_op=False):
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class SparsePayload(object):
    """Layout of the per-sample row every rank contributes to the embedding exchange:

        [ G : g_width floats | g_wide : 1 | pad to 4 | X : n_xcols floats | pad to 4 ]

    ``G`` first so that its base pointer and the row stride stay 16-byte aligned for float4 access."""

    def __init__(self, g_width, n_xcols):
        self.g_width = int(g_width)
        self.n_xcols = int(n_xcols)
        self.off_gw = self.g_width
        self.off_x = (self.g_width + 1 + 3) // 4 * 4
        self.ld = (self.off_x + self.n_xcols + 3) // 4 * 4

    def pack(self, X, G, g_wide):
        B = X.shape[0]
        row = torch.zeros((B, self.ld), dtype=torch.float32, device=X.device)
        if G is not None and self.g_width:
            row[:, :self.g_width] = G[:, :self.g_width]
        if g_wide is not None:
            row[:, self.off_gw] = g_wide
        row[:, self.off_x:self.off_x + self.n_xcols] = X[:, :self.n_xcols]
        return row

    def gather(self, row, group=None):
        world = dist.get_world_size(group)
        out = torch.empty((world * row.shape[0], self.ld), dtype=row.dtype, device=row.device)
        dist.all_gather_into_tensor(out, row, group=group)
        return out

    def views(self, gathered):
        """(X_all [NB, n_xcols] view, G_all [NB, g_width] view, g_wide_all [NB] contiguous)"""
        return (gathered[:, self.off_x:self.off_x + self.n_xcols], gathered[:, :self.g_width],
                gathered[:, self.off_gw].contiguous())


def fold_fm(g_out, width, out, fm_s, g_fm, emb_dim):
    """Row gradients with FM's backward folded in (interaction.py:26-34 under autograd):
    ``G[b, f, :] = g_out[b, f, :] + g_fm[b] * (S[b, :] - e[b, f, :])``.  Plain torch, any device."""
    B = out.shape[0] if out is not None else g_out.shape[0]
    G = g_out[:, :width] if g_out is not None else None
    if g_fm is not None:
        nf = width // emb_dim
        e = out[:, :width].reshape(B, nf, emb_dim)
        fold = (g_fm.reshape(B, 1, 1) * (fm_s[:, :emb_dim].unsqueeze(1) - e)).reshape(B, width)
        G = fold if G is None else G + fold
    return G


class DataParallelTrainer(object):
    """``trainer.train_step(xb, yb)`` == ``model._train_step`` on the concatenation of every rank's batch."""

    def __init__(self, model, process_group=None, broadcast_parameters=True):
        if not dist.is_initialized():
            raise RuntimeError("initialise torch.distributed first (backend 'nccl' = RCCL on ROCm)")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.plan = model.model_plan()
        if not self.plan.unit_path:
            raise NotImplementedError("data-parallel training needs fixed-length sparse features over distinct "
                                      "tables (the deterministic update kernel); pooled VarLen features are "
                                      "single-GPU for now")
        tables = set(id(p) for p in self.plan.table_params)
        self.bucket = DenseBucket([p for p in model.parameters() if id(p) not in tables])
        self.payload = SparsePayload(self.plan.emb_width, self.plan.n_xcols)
        if broadcast_parameters:
            with torch.no_grad():
                for p in model.parameters():
                    dist.broadcast(p.data, 0, group=process_group)
                for b in model.buffers():
                    dist.broadcast(b.data, 0, group=process_group)
        self._stash = None
        self.plan.exchange = self._defer          # EmbedFunction.backward hands its inputs over instead of updating

    def close(self):
        self.plan.exchange = None

    def _defer(self, **kw):
        if self._stash is not None:
            raise RuntimeError("two embedding backward passes in one data-parallel step are not supported")
        self._stash = kw

    def train_step(self, xb, yb):
        from ._hip import lib as L
        from ._hip.ops import _ptr
        model, plan = self.model, self.plan
        y_pred = model(xb).squeeze()
        model.optim.zero_grad()
        self.bucket.attach()
        if isinstance(model.loss_func, list):
            loss = sum([model.loss_func[i](y_pred[:, i], yb[:, i], reduction='sum') for i in range(model.num_tasks)])
        else:
            loss = model.loss_func(y_pred, yb.squeeze(), reduction='sum')
        total_loss = loss + model.get_regularization_loss() + model.aux_loss
        self._stash = None
        total_loss.backward()
        work = self.bucket.all_reduce(self.group, async_op=True)      # overlaps with the embedding exchange

        st = self._stash
        self._stash = None
        if st is not None:
            X = st["X"]
            G = fold_fm(st["g_out"], plan.emb_width, st["out"], st["fm_s"], st["g_fm"], plan.emb_dim) \
                if plan.deep else None
            gathered = self.payload.gather(self.payload.pack(X, G, st["g_wide"]), self.group)
            X_all, G_all, gw_all = self.payload.views(gathered)
            NB = gathered.shape[0]
            lib = L.lib()
            stream = L.stream_handle(X.device)
            kind = plan.update[0]
            if kind == "dense":
                plan.ensure_gacc()
                plan.prepare_dense_grads()
                opt, lr, eps = L.UPD_ACCUM, 0.0, 0.0
            elif kind in ("sgd", "sgd2"):
                opt, lr, eps = L.UPD_SGD, float(plan.update[1]), 0.0
            else:
                opt, lr, eps = L.UPD_ADAGRAD, float(plan.update[1]), float(plan.update[2])
            cplan = plan.bind(X.device)
            if not plan.update_kernel_ok(NB):
                raise RuntimeError("global batch %d is beyond the deterministic update kernel" % NB)
            ids_t = torch.empty((len(plan.units), NB), dtype=torch.int32, device=X.device)
            L.check(lib.dctr_embed_ids(plan.units_ptr(), len(plan.units), _ptr(X_all), gathered.stride(0), NB,
                                       _ptr(ids_t), stream), "dctr_embed_ids")
            L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), len(plan.units), plan.max_vocab, _ptr(ids_t), NB,
                                          _ptr(G_all) if plan.deep else None, gathered.stride(0), None, 0, None, 0,
                                          None, _ptr(gw_all) if plan.wide else None, opt, lr, eps, None, 0, None, stream),
                    "dctr_embed_update(global)")
        work.wait()
        model.optim.step()
        return loss.detach(), total_loss.detach(), y_pred.detach()
