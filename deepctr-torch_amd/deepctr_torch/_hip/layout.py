"""Where a table row, its optimizer state and the wide weight of the same id live in HBM.

Measured on MI355X (tools/micro/rowbench.hip, profiles/r02_rowbench.json): the memory system moves 128-byte lines --
a random 64-byte row (embedding_dim 16, fp32) costs what a random 128-byte row costs (48 vs 44 G rows/s), and a
read-modify-write of a row in one array plus its Adagrad state in another runs at 9.7 G rows/s against 17.5 G rows/s
for ONE 128-byte line holding both.  So whatever the update kernel touches together is stored together:

  "interleaved"  deep slab [V, 2D]: row r = [ e_r (D) | Adagrad sum_r (D) ]  -- one 128-byte line at D = 16;
                 wide slab [V, 2]:  row r = [ w_r | sum_r ]
  "block"        one slab per unit [V, 64]: row r = [ e_r (16) | w_r | wide sum_r | pad 14 | Adagrad sum_r (16) | pad 16 ]
                 -- the gather reads e_r and w_r from ONE line (26 instead of 52 random requests per sample)
  "contiguous"   the reference's layout: every tensor on its own
  "infer"        (apply_infer_layout, models that only ever predict) one slab per unit [V, 32]: row r = [ e_r (D <= 28) | w_r |
                 pad ] -- ONE 128-byte line per (sample, field) for the gather instead of two (the deep row's line and the wide
                 weight's line): profiles/r02_kernel_sweep_layouts.json, 302 against 391 us at B_eff 262 144

``nn.Embedding.weight`` and ``optimizer.state[p]['sum']`` become strided VIEWS of the slabs: ``state_dict`` keys,
shapes and values, ``optimizer.state_dict()``, ``load_state_dict`` and every torch op on them keep working; the
kernels read the row strides from the field descriptors (``dctr_field_t.ld`` / ``ld_state``)."""
import os

import torch

from .plan import _ParamMap

_SLAB = _ParamMap()       # table parameter -> (slab tensor, column of the parameter, column of its state | None)


def wanted_layout():
    return os.environ.get("DCTR_TABLE_LAYOUT", "interleaved")


def _seat(p, slab, col, state, scol):
    """Make ``p`` (and its state tensor) views of ``slab`` at the given columns, values preserved.  Returns the
    state view (or None)."""
    D = int(p.shape[1])
    cur = _SLAB.get(p)
    if cur is None or cur[0] is not slab:
        view = slab[:, col:col + D]
        view.copy_(p.data)
        p.data = view
        _SLAB[p] = (slab, col, scol)
    if state is None:
        return None
    sview = slab[:, scol:scol + D]
    if state.data_ptr() != sview.data_ptr() or state.stride() != sview.stride():
        sview.copy_(state)
    return sview


def _is_view_of(p, slab_entry):
    if slab_entry is None:
        return False
    slab, col, _ = slab_entry
    return slab.device == p.device and p.data_ptr() == slab.data_ptr() + 4 * col and p.stride(0) == slab.stride(0)


def apply_layout(plan, optimizer, state_key="sum", layout=None):
    """Re-seat the plan's tables (fixed-length fields over distinct tables: ``plan.unit_path``) and their ``state_key``
    optimizer state in the wanted layout.  Idempotent; a state tensor that was replaced since (``optimizer.
    load_state_dict``) is copied back into its slab.  Returns ``{param: state view}``."""
    layout = layout or wanted_layout()
    params = plan.table_params
    states = {id(p): optimizer.state[p][state_key] for p in params}
    if layout == "contiguous" or not plan.unit_path:
        return {p: states[id(p)] for p in params}
    out = {}
    for di, wi, _, _ in plan.units:
        pd = plan.deep[di].param if di >= 0 else None
        pw = plan.wide[wi].param if wi >= 0 else None
        block = layout == "block" and pd is not None and pw is not None and int(pd.shape[1]) == 16
        if block:
            ent = _SLAB.get(pd)
            slab = ent[0] if (_is_view_of(pd, ent) and ent[0].shape[1] == 64 and _is_view_of(pw, _SLAB.get(pw)) and
                              _SLAB.get(pw)[0] is ent[0]) else \
                torch.zeros((int(pd.shape[0]), 64), dtype=torch.float32, device=pd.device)
            for p, col, scol in ((pd, 0, 32), (pw, 16, 17)):
                sv = _seat(p, slab, col, states[id(p)], scol)
                optimizer.state[p][state_key] = sv
                out[p] = sv
            continue
        for p in (pd, pw):
            if p is None:
                continue
            D = int(p.shape[1])
            ent = _SLAB.get(p)
            # (the slab a parameter already lives in is reused only if it is THIS layout's: same width AND the state column
            # recorded for it -- a forward-only [V, 32] slab of a D = 16 table has the width of the interleaved one but
            # keeps the wide weight where the Adagrad state would go)
            slab = ent[0] if (_is_view_of(p, ent) and ent[0].shape[1] == 2 * D and ent[2] == D) else \
                torch.empty((int(p.shape[0]), 2 * D), dtype=torch.float32, device=p.device)
            sv = _seat(p, slab, 0, states[id(p)], D)
            optimizer.state[p][state_key] = sv
            out[p] = sv
    return out


def apply_infer_layout(plan):
    """Forward-only seating (round 5): a model that was never compiled for training -- ``predict()`` / ``evaluate()`` on
    loaded weights -- pays two random 128-byte lines per (sample, field) in the reference's layout, the deep row's and the
    wide weight's.  Every unit with a deep table of D <= 28 floats (a multiple of 4) and a wide table over the same ids is
    re-seated in ONE ``[V, 32]`` slab, row r = ``[ e_r | w_r | pad ]``; ``weight`` tensors become strided views (``state_dict``
    keys, shapes and values unchanged).  Idempotent; ``compile()`` re-seats again for the optimizer it gets (apply_layout).
    Returns the number of units re-seated by this call."""
    if os.environ.get("DCTR_PREDICT_LAYOUT", "1") == "0":
        return 0
    n = 0
    for di, wi, _, _ in plan.units:
        if di < 0 or wi < 0:
            continue
        pd, pw = plan.deep[di].param, plan.wide[wi].param
        D = int(pd.shape[1])
        if D > 28 or D % 4 or int(pw.shape[1]) != 1 or pd.shape[0] != pw.shape[0] or pd.device != pw.device:
            continue
        ed, ew = _SLAB.get(pd), _SLAB.get(pw)
        if _is_view_of(pd, ed) and _is_view_of(pw, ew) and ed[0] is ew[0] and ed[0].shape[1] == 32:
            continue
        if any(_is_view_of(p, e) for p, e in ((pd, ed), (pw, ew))):
            continue        # seated for training (interleaved / block): not a forward-only model, leave it
        slab = torch.zeros((int(pd.shape[0]), 32), dtype=torch.float32, device=pd.device)
        with torch.no_grad():
            _seat(pd, slab, 0, None, None)
            _seat(pw, slab, D, None, None)
        n += 1
    return n
