"""``model.fit()`` under ``torchrun`` (WORLD_SIZE > 1): one process per GPU, the minibatch sharded over the ranks.

Reference: ``basemodel.py:206-209`` -- with ``gpus`` the reference wraps the model in ``nn.DataParallel`` and sets
``batch_size *= len(gpus)``: the caller's ``batch_size`` is PER GPU, a step trains on ``batch_size x n_gpus`` samples, the
loss is a sum over that global batch (``:254``), metrics are computed on the gathered predictions (``:264-269``).  Here the
same contract, one process per GPU (``torch.distributed``: RCCL on GPUs, gloo on CPU stand-ins):

  * every rank holds the whole dataset and draws the reference's epoch permutation (the same draws from the default
    generator; rank 0's permutation is broadcast so that a rank seeded differently cannot fork the run);
  * step k trains on rows ``order[k G : (k + 1) G]`` of the permutation, ``G = batch_size x world``; rank r contributes the
    slice ``[r b, (r + 1) b)`` of them; the step itself is ``parallel.ShardedTrainer.train_step`` (tables sharded over the
    ranks, tower data-parallel): one optimizer step on the gradient of the global batch -- the single-process step on G
    samples;
  * a ragged last batch (fewer than G rows) does not split evenly: the owners' tables are made current everywhere
    (``gather_tables``) and EVERY rank runs the single-process step on the whole ragged batch -- identical inputs and
    deterministic kernels leave identical replicas, owners included;
  * ``History``: the epoch loss is the all-reduced sum over every sample / sample count; every metric is evaluated per
    GLOBAL batch on the all-gathered (label, prediction) pairs, as ``DataParallel`` hands them to the reference;
  * validation, callbacks and whatever reads ``model.embedding_dict`` after an epoch see complete tables
    (``gather_tables`` at the end of every epoch that needs it).

Pooled VarLen features and tables shared through ``embedding_name`` are sharded too (round 6: the owner of a table group pools
its positions locally and ships one row per field).  Models / optimizers outside ``ShardedTrainer``'s envelope (Adam / RMSprop,
L2 on the tables -- the reference's default kwargs --, unequal embedding_dims) train through ``parallel.DataParallelTrainer`` instead: tables
replicated, every rank applies the same global update from the all-gathered row gradients -- the single-process step on the
global batch as well, O(vocabulary) only where the optimizer itself is.  Never a silent single-GPU run."""
import os
import time

import numpy as np
import torch


def context():
    """(world, rank) when this process is one of several training ranks, else None.  ``DCTR_FIT_FORCE_TRAINER=1``: also for a
    world of ONE rank -- fit() then runs through the sharded trainer and its exchange with nobody to exchange with (what
    bench.py's ``fit_api_sharded_1rank`` leg and the one-GPU tests time and check)."""
    import torch.distributed as dist
    force = os.environ.get("DCTR_FIT_FORCE_TRAINER") == "1"
    if dist.is_available() and dist.is_initialized():
        w = dist.get_world_size()
        return (w, dist.get_rank()) if (w > 1 or force) else None
    if force and os.environ.get("DCTR_FIT_DISTRIBUTED", "1") != "0":
        os.environ.setdefault("WORLD_SIZE", "1")
        return (int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("DCTR_FIT_DISTRIBUTED", "1") != "0":
        return (int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")))
    return None


def _ensure_group(device):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        cuda = torch.device(device).type == "cuda"
        kw = dict(device_id=torch.device(device)) if cuda else {}
        dist.init_process_group("nccl" if cuda else "gloo", rank=int(os.environ.get("RANK", "0")),
                                world_size=int(os.environ["WORLD_SIZE"]), **kw)
    return dist


class _Sharded(object):
    """ShardedTrainer behind the few calls the epoch loop makes."""
    kind = "tables sharded over the ranks"

    def __init__(self, tr):
        self.tr = tr
        self.exchange = tr.exchange

    def train_step(self, xb, yb, next_xb=None):
        return self.tr.train_step(xb, yb, next_xb=next_xb)

    def train_block(self, x_block, y_block, next_first=None):
        """S consecutive steps as one hipGraph over the direct exchange (ShardedTrainer.train_block) -> this rank's data loss
        summed over the S steps (fp64 device scalar)"""
        self.tr.train_block(x_block, y_block, next_first=next_first)
        return self.tr.last_block_loss

    def blocks(self, dev):
        """steps per hipGraph for full groups of batches, 0: step by step"""
        if self.tr.exchange != "direct" or dev.type != "cuda" or self.tr.slab is None or \
                os.environ.get("DCTR_FIT_GRAPH", "1") == "0":
            return 0
        return max(0, int(os.environ.get("DCTR_FIT_STEPS_PER_GRAPH", "16")))

    def set_use_graphs(self, on):
        self.tr.set_use_graphs(on)

    def graphs_on(self):
        """an earlier fit() call left the trainer replaying captured segments: this call starts on them"""
        return bool(self.tr.use_graphs)

    def gather_tables(self):
        self.tr.gather_tables()

    def detach(self, train=False):
        self.tr._join()
        self.tr.plan.sharder = None

    def attach(self):
        self.tr.plan.sharder = self.tr
        self.tr._announced = None      # whatever was announced to the owners is stale now
        self.tr._pre = None


class _Replicated(object):
    """DataParallelTrainer (replicated tables, any model / optimizer) behind the same calls: every replica's tables are
    always current, nothing to gather, nothing to announce."""
    kind = "tables replicated"
    exchange = "all-gather of row gradients"

    def __init__(self, tr):
        self.tr = tr

    def train_step(self, xb, yb, next_xb=None):
        return self.tr.train_step(xb, yb)

    def blocks(self, dev):
        return 0

    def set_use_graphs(self, on):
        pass

    def graphs_on(self):
        return False

    def gather_tables(self):
        pass

    def detach(self, train=False):
        # (train: a ragged batch taken by every rank on its own runs the autograd route + torch.optim like the trainer's
        # steps do -- one owner of the dense optimizer state (Adam's step counters) whichever step runs; the switch is put
        # back by attach())
        self.tr.plan.exchange = None
        self._env = None
        if train:
            self._env = (os.environ.get("DCTR_FUSED_STEP"),)
            os.environ["DCTR_FUSED_STEP"] = "0"

    def attach(self):
        self.tr.plan.exchange = self.tr._defer
        env = getattr(self, "_env", None)
        if env is not None:
            if env[0] is None:
                os.environ.pop("DCTR_FUSED_STEP", None)
            else:
                os.environ["DCTR_FUSED_STEP"] = env[0]
            self._env = None


def trainer_for(model):
    """The model's trainer (built once per compile(): it holds exchange buffers and captured segments): ShardedTrainer where
    its envelope holds the model, DataParallelTrainer otherwise."""
    from . import parallel as par
    ad = getattr(model, "_dist_trainer", None)
    plan = model.model_plan()
    if ad is not None and not getattr(ad, "_closed", False) and ad.tr.model is model and \
            getattr(ad, "_optim", None) is getattr(model, "optim", None) and ad.tr.plan is plan:
        ad.attach()                # (the previous fit() left every rank's tables current and detached the trainer)
        return ad
    ad = None
    factory = getattr(model, "_shard_ops_factory", None)      # (tests: stand-ins for the device kernels, simple plans only)
    if (plan.simple_units or factory is None) and plan.unit_path and plan.update[0] in ("sgd", "adagrad") and \
            os.environ.get("DCTR_FIT_TRAINER", "auto") != "replicated":
        try:
            ops = None
            if factory is not None:
                ops = factory(model, par.ShardLayout(plan, *context()))
            # 'auto': the direct exchange (whole steps, exchanges included, as hipGraphs of S steps) wherever its
            # self-test passes on every rank, RCCL otherwise -- parallel.resolve_exchange says which and why
            ad = _Sharded(par.ShardedTrainer(model, ops=ops, exchange=os.environ.get("DCTR_SHARDED_EXCHANGE", "auto")))
        except NotImplementedError:
            plan.sharder = None
            ad = None
    if ad is None:
        ad = _Replicated(par.DataParallelTrainer(model))
    ad._optim = getattr(model, "optim", None)
    model._dist_trainer = ad
    return ad


class _Unsharded(object):
    """``with _Unsharded(trainer):`` -- lookups and updates go through the local tables (current after ``gather_tables``)."""

    def __init__(self, tr, train=False):
        self.tr, self.train = tr, train

    def __enter__(self):
        self.tr.detach(train=self.train)
        return self

    def __exit__(self, *exc):
        self.tr.attach()
        return False


def fit(model, X_all, y_all, batch_size, epochs, verbose, initial_epoch, do_validation, val_x, val_y, shuffle, callbacks):
    from . import callbacks as _cb
    dist = _ensure_group(model.device)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = X_all.device
    tr = trainer_for(model)
    b, G = int(batch_size), int(batch_size) * world
    sample_num = X_all.shape[0]
    n_full, n_tail = sample_num // G, sample_num % G
    steps_per_epoch = n_full + (1 if n_tail else 0)
    model.train()
    if rank == 0:
        note = getattr(getattr(tr, "tr", None), "exchange_note", "")
        print("%s -- %d ranks, batch %d per rank (%d global), %s (%s exchange%s)" % (
            model.device, world, b, G, tr.kind, tr.exchange, ": " + note if note else ""))
        print("Train on {0} samples, validate on {1} samples, {2} steps per epoch".format(
            sample_num, len(val_y), steps_per_epoch))
    # every rank runs the callbacks (EarlyStopping must see the logs everywhere: its decision is all-reduced below), but a
    # callback that WRITES -- ModelCheckpoint -- does so on rank 0 only: the ranks hold the same gathered parameters, and
    # several processes pickling into one path at once can leave a truncated file (round-5 advisor finding)
    if callbacks and rank != 0:
        callbacks = [c for c in callbacks if not isinstance(c, _cb.ModelCheckpoint)]
    cbs = _cb.CallbackList((callbacks or []) + [model.history])
    cbs.set_model(model)
    cbs.on_train_begin()
    cbs.set_model(model)
    model.stop_training = False
    plan = model.model_plan()

    def draw_order():
        # the reference's two draws per epoch (see BaseModel.fit); rank 0's permutation wins
        torch.empty((), dtype=torch.int64).random_()
        if not shuffle:
            return None
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        gen = torch.Generator()
        gen.manual_seed(seed)
        order = torch.randperm(sample_num, generator=gen).to(dev)
        dist.broadcast(order, 0)
        return order

    def rows(lo, hi):
        if order is not None:
            idx = order[lo:hi]
            return X_all.index_select(0, idx), y_all.index_select(0, idx)
        return X_all[lo:hi].contiguous(), y_all[lo:hi].contiguous()

    graphs_on = tr.graphs_on()
    S_blk = tr.blocks(dev)
    if S_blk < 2:
        S_blk = 0
    blk_rows = None
    if S_blk:
        # row positions of S consecutive steps' slices of THIS rank inside the epoch order: step j of a block takes
        # order[(k + j) G + r b : ... + b] -- one index_select per block instead of one per step
        blk_rows = (torch.arange(S_blk, device=dev)[:, None] * G + rank * b + torch.arange(b, device=dev)[None, :]).reshape(-1)

    def block_rows(step):
        pos = blk_rows + step * G
        idx = order.index_select(0, pos) if order is not None else pos
        return (X_all.index_select(0, idx).view(S_blk, b, X_all.shape[1]), y_all.index_select(0, idx).view((S_blk, b) + tuple(y_all.shape[1:])))

    for epoch in range(initial_epoch, epochs):
        cbs.on_epoch_begin(epoch)
        epoch_logs = {}
        start_time = time.time()
        order = draw_order()
        # this rank's share of the full batches: the DATA loss (all-reduced below) and, apart, what each step adds that does
        # not depend on the samples -- regularisation and auxiliary terms, which every rank computes in full on identical
        # dense replicas: they enter the epoch's loss ONCE per step, not once per rank (round-5 advisor finding)
        acc_sharded = torch.zeros((), device=dev, dtype=torch.float64)
        acc_terms = torch.zeros((), device=dev, dtype=torch.float64)
        acc_tail = torch.zeros((), device=dev, dtype=torch.float64)       # the ragged batch (the same on every rank)
        preds = [] if (verbose > 0 and model.metrics) else None
        current = False                 # every rank's tables current?
        nxt = rows(rank * b, (rank + 1) * b) if n_full else None
        step, blk_next = 0, None
        while step < n_full:
            model._sync_optimizer_hyper(full=False)      # (a callback may have moved lr: the single-process loop looks too)
            if not graphs_on and step >= 2 and dev.type == "cuda" and os.environ.get("DCTR_FIT_GRAPH", "1") != "0":
                tr.set_use_graphs(True)     # (two eager steps first: descriptors are uploaded, buffers allocated)
                graphs_on = True
            if S_blk and graphs_on and preds is None and step + S_blk <= n_full:
                # S steps as ONE hipGraph, exchanges included (ShardedTrainer.train_block over the direct exchange).  What
                # follows is announced with the last step's gradients -- the NEXT block's own first batch when another full
                # block follows (the trainer recognises an announced batch by its storage), else the next single batch
                xs, ys = blk_next if blk_next is not None else block_rows(step)
                blk_next, nf = None, None
                if step + 2 * S_blk <= n_full:
                    blk_next = block_rows(step + S_blk)
                    nf = (blk_next[0][0], blk_next[1][0])
                elif step + S_blk < n_full:
                    lo = (step + S_blk) * G + rank * b
                    nf = rows(lo, lo + b)
                acc_sharded += tr.train_block(xs, ys, next_first=nf[0] if nf is not None else None)
                step += S_blk
                nxt = nf
                continue
            xb, yb = nxt if nxt is not None else rows(step * G + rank * b, step * G + (rank + 1) * b)
            lo = (step + 1) * G + rank * b
            nxt = rows(lo, lo + b) if step + 1 < n_full else None
            loss, total_loss, y_pred = tr.train_step(xb, yb, next_xb=nxt[0] if nxt is not None else None)
            data = loss.detach().double().sum()
            acc_sharded += data
            acc_terms += total_loss.detach().double().sum() - data
            if preds is not None:
                preds.append((yb, y_pred.detach().reshape(-1).clone(), True))
            step += 1
        if n_tail:
            tr.gather_tables()
            with _Unsharded(tr, train=True):
                xb, yb = rows(n_full * G, sample_num)
                loss, total_loss, y_pred = model._train_step(xb, yb)
            acc_tail += total_loss.detach().double().sum()
            current = True
            if preds is not None:
                preds.append((yb, y_pred.detach().reshape(-1).clone(), False))
        dist.all_reduce(acc_sharded)
        plan.check_ids()
        epoch_logs["loss"] = float((acc_sharded + acc_terms + acc_tail).item()) / sample_num
        if preds is not None:
            # the reference's metric of every (global) batch, averaged over steps (basemodel.py:264-269,280)
            per_batch = {name: [] for name in model.metrics}
            for yb, yp, sharded in preds:
                if sharded:
                    both = torch.stack([yb.reshape(-1).float(), yp.float()])
                    parts = [torch.empty_like(both) for _ in range(world)]
                    dist.all_gather(parts, both)
                    both = torch.cat(parts, dim=1)
                    yt, ypn = both[0].cpu().numpy(), both[1].cpu().numpy().astype("float64")
                else:
                    yt, ypn = yb.cpu().numpy(), yp.cpu().numpy().astype("float64")
                for name, fun in model.metrics.items():
                    per_batch[name].append(fun(yt, ypn))
            for name, vals in per_batch.items():
                epoch_logs[name] = np.sum(vals) / steps_per_epoch
        last = epoch + 1 >= epochs
        if (do_validation or callbacks or last) and not current:
            tr.gather_tables()
            current = True
        if do_validation:
            with _Unsharded(tr):
                for name, result in model.evaluate(val_x, val_y, batch_size).items():
                    epoch_logs["val_" + name] = result
            model.train()
        if verbose > 0 and rank == 0:
            epoch_time = int(time.time() - start_time)
            print('Epoch {0}/{1}'.format(epoch + 1, epochs))
            eval_str = "{0}s - loss: {1: .4f}".format(epoch_time, epoch_logs["loss"])
            for name in model.metrics:
                eval_str += " - " + name + ": {0: .4f}".format(epoch_logs[name])
            if do_validation:
                for name in model.metrics:
                    eval_str += " - " + "val_" + name + ": {0: .4f}".format(epoch_logs["val_" + name])
            print(eval_str)
        if callbacks:
            with _Unsharded(tr):
                cbs.on_epoch_end(epoch, epoch_logs)
            dist.barrier()         # (rank 0 may have written a checkpoint: nobody runs ahead of it)
        else:
            cbs.on_epoch_end(epoch, epoch_logs)
        stop = torch.tensor([1 if model.stop_training else 0], device=dev)
        dist.all_reduce(stop, op=dist.ReduceOp.MAX)       # (EarlyStopping must stop every rank: the steps are collective)
        if int(stop.item()):
            model.stop_training = True
            if not current:
                tr.gather_tables()
            break
    # leave the model usable on its own (predict / evaluate / state_dict read complete local tables); the trainer is kept
    # for the next fit() call and re-attached there
    tr.detach()
    if isinstance(tr, _Replicated):
        # hand the model back as it was: a later single-process fit() takes the lazy O(batch) update again (the trainer
        # forced the dense route); the next distributed fit() builds a new trainer
        tr.tr.close()
        tr._closed = True
    cbs.on_train_end()
    return model.history
