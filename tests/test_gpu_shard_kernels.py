"""GPU: the device steps of the table-sharded exchange for world sizes > 1, WITHOUT any communication.

The RCCL trainer has only ever run at one rank on hardware (one GPU per box), and the gloo tests replace its kernels by
the torch stand-in (tests/shard_standin.py).  That left the N > 1 code paths of the HIP kernels themselves -- slot
layout of the chunks, the owners' gather over the global batch, ``dctr_shard_assemble_fwd / _bwd``, the owners' update
from ``N * B`` gradient rows -- unexecuted.  Here every rank's ``HipShardOps`` of a world of 2, 3 and 8 is run on one GPU
on synthetic "received" buffers and compared with the stand-in on a CPU copy of the model (same layouts by
construction: both read ``ShardLayout``).  Tolerances: gathers / scatters of rows are exact; sums (wide logit, FM, the
update's duplicate-row sums) 2e-6 relative."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import build_model, load_golden

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from shard_standin import TorchShardOps  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, tol=2e-6, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol * max(1.0, float(b.abs().max()) if b.numel() else 1.0), "%s: %.3e" % (what, err)


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shard_kernels_of_every_rank_match_the_standin(world, opt):
    from deepctr_torch import parallel as par
    g = load_golden("deepfm_criteo")
    models = []
    for dev in (DEV, "cpu"):
        m = build_model(g["spec"], dev)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        models.append(m)
    mg, mc = models
    pg, pc = mg.model_plan(), mc.model_plan()
    pg.bind(torch.device(DEV))
    assert pg.update[0] == opt and pc.update[0] == opt
    B = 37                                        # odd: ragged tiles everywhere
    gen = torch.Generator().manual_seed(world)
    Xc = torch.from_numpy(g["X"])[:B].clone()
    while Xc.shape[0] < B:
        Xc = torch.cat([Xc, Xc], 0)[:B]
    Xg = Xc.to(DEV)
    for rank in range(world):
        lg, lc = par.ShardLayout(pg, world, rank), par.ShardLayout(pc, world, rank)
        og, oc = par.HipShardOps(mg, lg), TorchShardOps(mc, lc)
        assert (lg.ldc, lg.n_slots, lg.wide_col, lg.owned) == (lc.ldc, lc.n_slots, lc.wide_col, lc.owned)
        # ids every owner needs of this rank's samples
        _close(og.pack_ids(Xg).contiguous(), oc.pack_ids(Xc).contiguous(), 0.0, "pack_ids")
        # the owner's gather over the global batch (ids of all N*B samples for MY slots)
        NB = world * B
        vocab = [int(pc.deep[pc.units[u][0]].vocab) for u in lc.owned]
        ids_all = torch.zeros(NB, lc.n_slots)
        for j, v in enumerate(vocab):
            ids_all[:, j] = torch.randint(0, v, (NB,), generator=gen).float()
        ch_g, ids_g = og.gather(ids_all.to(DEV))
        ch_c, ids_c = oc.gather(ids_all)
        nown = len(lc.owned)
        if nown:
            _close(ch_g[:, :nown * lc.D], ch_c[:, :nown * lc.D], 0.0, "gathered rows")
            if lc.has_wide:
                _close(ch_g[:, lc.wide_col], ch_c[:, lc.wide_col], 2e-6, "wide partial")
            assert torch.equal(ids_g[0].cpu(), ids_c)
        # what arrives from the owners -> the single-GPU outputs of the lookup
        recv = torch.randn(NB, lc.ldc, generator=gen)
        out_g, wide_g, fm_g, fms_g = og.assemble_fwd(recv.to(DEV), Xg, True)
        out_c, wide_c, fm_c, fms_c = oc.assemble_fwd(recv, Xc, True)
        W = lc.F * lc.D
        _close(out_g[:, :W], out_c[:, :W], 0.0, "assembled rows")
        if pc.dense_cols:
            _close(out_g[:, pc.dense_off:pc.dense_off + len(pc.dense_cols)],
                   out_c[:, pc.dense_off:pc.dense_off + len(pc.dense_cols)], 0.0, "dense block")
        if wide_c is not None:
            _close(wide_g, wide_c, 2e-6, "wide logit")
        _close(fm_g, fm_c, 2e-5, "fm")
        _close(fms_g[:, :lc.D], fms_c, 2e-6, "fm_s")
        # gradients of the lookup outputs -> what travels back to the owners
        g_out = torch.randn(B, pc.ld_out, generator=gen)
        g_wide = torch.randn(B, generator=gen) if pc.has_wide else None
        g_fm = torch.randn(B, generator=gen)
        gwd_g = torch.empty(len(pc.wdense_cols), 1, device=DEV) if pc.wide_dense_weight is not None else None
        gwd_c = torch.empty(len(pc.wdense_cols), 1) if pc.wide_dense_weight is not None else None
        send_g = og.assemble_bwd(Xg, g_out.to(DEV), g_wide.to(DEV) if g_wide is not None else None, g_fm.to(DEV),
                                 out_g, fms_g, gwd_g).view(world, B, lc.ldc)
        send_c = oc.assemble_bwd(Xc, g_out, g_wide, g_fm, out_c, fms_c, gwd_c).view(world, B, lc.ldc)
        for f in range(lc.F):
            q, j = lc.owner[f], lc.slot[f]
            _close(send_g[q, :, j * lc.D:(j + 1) * lc.D], send_c[q, :, j * lc.D:(j + 1) * lc.D], 2e-6,
                   "row gradient of unit %d" % f)
        if lc.has_wide:
            _close(send_g[:, :, lc.wide_col], send_c[:, :, lc.wide_col], 0.0, "wide gradient")
        if gwd_c is not None:
            _close(gwd_g, gwd_c, 2e-5, "g Linear.weight")
        # the owner's update from the N*B gradient rows of the global batch
        if nown:
            grads_all = torch.randn(NB, lc.ldc, generator=gen) * 0.1
            og.update(grads_all.to(DEV), ids_g)
            oc.update(grads_all, ids_c)
            torch.cuda.synchronize()
            for u in lc.owned:
                di, wi = pc.units[u][0], pc.units[u][1]
                _close(pg.deep[di].param, pc.deep[di].param, 2e-6, "deep table of unit %d" % u)
                if wi >= 0:
                    _close(pg.wide[wi].param, pc.wide[wi].param, 2e-6, "wide table of unit %d" % u)
    pg.check_ids()
