"""torch stand-ins for the four device steps of the table-sharded exchange (HipShardOps in deepctr_torch/parallel.py),
for the CPU / gloo tests of the exchange logic.  Test infrastructure only: same layouts, plain torch index ops."""
import torch


class TorchShardOps(object):
    def __init__(self, model, layout):
        self.lay, self.plan = layout, model.model_plan()
        plan = self.plan
        self.fields = []                                   # (deep param, wide param | None) per owned slot
        for u in layout.owned:
            di, wi, col, _ = plan.units[u]
            self.fields.append((plan.deep[di].param, plan.wide[wi].param if wi >= 0 else None))

    def pack_ids(self, X):
        return self.lay.pack_ids(X, torch.tensor(self.lay.id_cols, dtype=torch.long))

    def gather(self, ids_all):
        lay = self.lay
        NB = ids_all.shape[0]
        chunks = torch.full((NB, lay.ldc), float("nan"))        # unused slots must never be read
        ids = ids_all.long()
        wide = torch.zeros(NB)
        with torch.no_grad():
            for j, (dp, wp) in enumerate(self.fields):
                chunks[:, j * lay.D:(j + 1) * lay.D] = dp[ids[:, j]]
                if wp is not None:
                    wide += wp[ids[:, j], 0]
        if lay.has_wide:
            chunks[:, lay.wide_col] = wide
        return chunks, ids[:, :len(self.fields)].t().contiguous().int()

    def assemble_fwd(self, recv, X, want_fm):
        lay, plan = self.lay, self.plan
        B = X.shape[0]
        out = torch.zeros(B, plan.ld_out)
        r = recv.view(lay.world, B, lay.ldc)
        for f in range(lay.F):
            q, j = lay.owner[f], lay.slot[f]
            out[:, f * lay.D:(f + 1) * lay.D] = r[q, :, j * lay.D:(j + 1) * lay.D]
        if plan.dense_cols:
            out[:, plan.dense_off:plan.dense_off + len(plan.dense_cols)] = X[:, plan.dense_cols]
        wide = None
        if plan.has_wide:
            wide = torch.zeros(B)
            if lay.has_wide:
                for q in range(min(lay.world, lay.F)):
                    wide = wide + r[q, :, lay.wide_col]
            if plan.wide_dense_weight is not None:
                wide = wide + (X[:, plan.wdense_cols] @ plan.wide_dense_weight.detach()).squeeze(1)
        fm = fm_s = None
        if want_fm:
            E = out[:, :lay.F * lay.D].view(B, lay.F, lay.D)
            fm_s = E.sum(1)
            fm = 0.5 * (fm_s.pow(2) - E.pow(2).sum(1)).sum(1)
        return out, wide, fm, fm_s

    def assemble_bwd(self, X, g_out, g_wide, g_fm, out, fm_s, g_wdense):
        lay, plan = self.lay, self.plan
        B = X.shape[0]
        send = torch.full((lay.world, B, lay.ldc), float("nan"))
        W = lay.F * lay.D
        G = g_out[:, :W].clone() if g_out is not None else torch.zeros(B, W)
        if g_fm is not None:
            E = out[:, :W].view(B, lay.F, lay.D)
            G = G + (g_fm.view(B, 1, 1) * (fm_s[:, :lay.D].unsqueeze(1) - E)).reshape(B, W)
        for f in range(lay.F):
            q, j = lay.owner[f], lay.slot[f]
            send[q, :, j * lay.D:(j + 1) * lay.D] = G[:, f * lay.D:(f + 1) * lay.D]
        if lay.has_wide:
            send[:, :, lay.wide_col] = (g_wide if g_wide is not None else torch.zeros(B)).unsqueeze(0)
        if g_wdense is not None:
            g_wdense.copy_((X[:, plan.wdense_cols].t() @ g_wide).view_as(g_wdense))
        return send.view(lay.world * B, lay.ldc)

    def update(self, grads_all, ids_t):
        from deepctr_torch._hip.plan import _STATE
        lay, plan = self.lay, self.plan
        kind = plan.update[0]
        lr = float(plan.update[1])
        eps = float(plan.update[2]) if kind == "adagrad" else 0.0
        with torch.no_grad():
            for j, (dp, wp) in enumerate(self.fields):
                ids = ids_t[j].long()
                for tbl, g in ((dp, grads_all[:, j * lay.D:(j + 1) * lay.D]),
                               (wp, grads_all[:, lay.wide_col:lay.wide_col + 1] if wp is not None else None)):
                    if tbl is None:
                        continue
                    Gd = torch.zeros_like(tbl).index_add_(0, ids, g.contiguous())
                    if kind == "sgd":
                        tbl -= lr * Gd
                    else:
                        st = _STATE[tbl]
                        st += Gd * Gd
                        tbl -= lr * Gd / (st.sqrt() + eps)
