"""CPU: feature-column API, X layout, plan compilation, state_dict keys, update-mode selection, pickling."""
import io

import numpy as np
import pytest
import torch

from helpers import build_model, feature_columns, golden_names, load_golden
from np_oracle import build_input_features as oracle_layout


def test_feature_column_api():
    from deepctr_torch.inputs import (DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features,
                                      get_feature_names)
    s = SparseFeat("c", 100, "auto")
    assert s.embedding_dim == 6 * int(pow(100, 0.25)) and s.embedding_name == "c" and s.group_name == "default_group"
    assert SparseFeat("c", 3) == SparseFeat("c", 3) and hash(SparseFeat("c", 3)) == hash("c")
    v = VarLenSparseFeat(SparseFeat("h", 10, 8, embedding_name="item"), maxlen=4, combiner="sum", length_name="hl")
    assert (v.name, v.vocabulary_size, v.embedding_dim, v.embedding_name, v.maxlen) == ("h", 10, 8, "item", 4)
    d = DenseFeat("p", 3)
    cols = [s, d, v, SparseFeat("c", 100, 8)]                     # duplicate name: first wins
    assert build_input_features(cols) == {"c": (0, 1), "p": (1, 4), "h": (4, 8), "hl": (8, 9)}
    assert get_feature_names(cols) == ["c", "p", "h", "hl"]
    with pytest.raises(TypeError):
        build_input_features([object()])


@pytest.mark.parametrize("name", golden_names())
def test_layout_and_state_dict_keys_match_reference(name):
    g = load_golden(name)
    spec = g["spec"]
    model = build_model(spec, "cpu")
    assert dict(model.feature_index) == dict(oracle_layout(spec["linear_columns"] + spec["dnn_columns"]))
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(g["params"].keys())
    for k, v in g["params"].items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()})


def test_plan_layout_is_combined_dnn_input_order():
    g = load_golden("deepfm_mixed")
    model = build_model(g["spec"], "cpu")
    plan = model.model_plan()
    names = [f.name for f in plan.deep]
    assert names == ["user", "item", "cate", "hist_sum", "tags_mean", "kw_max", "seq_len_mean"]
    assert [f.out_off for f in plan.deep] == [0, 4, 8, 12, 16, 20, 24]
    assert plan.n_deep_fixed == 3 and plan.emb_dim == 4 and plan.vec == 4 and plan.has_maxpool
    assert plan.dense_off == 28 and plan.width == 32 and plan.ld_out % 4 == 0
    hist = plan.deep[3]
    assert hist.param is model.embedding_dict["item"].weight and hist.pool == 1 and hist.len == 4
    lenf = plan.deep[6]
    assert lenf.len_col == model.feature_index["seq_len_mean_length"][0] and lenf.pool == 2
    assert len(plan.table_params) == 6 + 6          # `item` is shared by two fields on both sides


def test_update_mode_selection():
    g = load_golden("deepfm_criteo")
    m = build_model(g["spec"], "cpu", l2=0.0)
    m.model_plan()
    m.compile("sgd", "binary_crossentropy")
    assert m._plan.update == ("sgd", 0.01)
    m.compile("adagrad", "binary_crossentropy")
    assert m._plan.update[0] == "adagrad" and m._plan.update[1:] == (0.01, 1e-10)
    m.compile("adam", "binary_crossentropy")          # every row moves every step: the exact lazy replay
    assert m._plan.update == ("lazy", "adam") and m._plan.lazy.kind == "adam"
    assert all("exp_avg" in m.optim.state[p] for p in m._plan.table_params)
    m.compile(torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9), "binary_crossentropy")
    assert m._plan.update == ("dense",) and m._plan.lazy is None
    m2 = build_model(g["spec"], "cpu", l2=1e-5)       # reference default: an L2 gradient on every row of every table
    m2.model_plan()
    m2.compile("adagrad", "binary_crossentropy")
    assert m2._plan.update == ("lazy", "adagrad")
    assert set(m2._plan.lazy.l2.values()) == {1e-5}
    m2.compile("sgd", "binary_crossentropy")
    assert m2._plan.update == ("lazy", "sgd")


def test_lazy_update_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("DCTR_LAZY_UPDATE", "0")
    g = load_golden("deepfm_criteo")
    m = build_model(g["spec"], "cpu", l2=1e-5)
    m.model_plan()
    m.compile("adam", "binary_crossentropy")
    assert m._plan.update == ("dense",)


def test_model_pickles_with_plan():
    g = load_golden("deepfm_mixed")
    m = build_model(g["spec"], "cpu")
    m.model_plan()
    m.compile("adagrad", "binary_crossentropy")
    buf = io.BytesIO()
    torch.save(m, buf)
    m2 = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)
    assert sorted(m2.state_dict()) == sorted(m.state_dict())
    assert m2._plan.cplan.n_deep == 0 and m2._plan.width == m._plan.width   # device image re-baked lazily


def test_fit_rejects_nothing_silently_on_cpu():
    g = load_golden("deepfm_fm_only")
    m = build_model(g["spec"], "cpu")
    m.compile("sgd", "binary_crossentropy")
    x = {c["name"]: g["X"][:, i] for i, c in enumerate(g["spec"]["dnn_columns"])}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.fit(x, g["y"], batch_size=8, epochs=1, verbose=0)


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r02_bench_driver_flags.json is a verbatim bench.py line from an MI355X run (--steps 20 --warmup 5, the
    driver's flags): the keys the driver and the judge read must all be there (a reminder to keep bench.py's output shape
    stable), every timed step a graph replay, and the other two 1-GPU configurations of BASELINE.json beside the headline."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_bench_driver_flags.json")
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["roofline"]["bound"] in ("hbm", "mfma") and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["config"]["eager_steps_in_timed_region"] == 0
    for name in ("xdeepfm", "fibinet"):
        leg = d["other_configs"][name]
        assert leg["roofline"]["bound"] == "mfma" and leg["hip_graph"] and leg["ms_per_step"] > 0
    assert d["roofline"]["traffic"] == d["roofline"]["traffic_detail"]["fetch_raw"] + d["roofline"]["traffic_detail"]["write_raw"]


@pytest.mark.parametrize("F", [2, 3, 5, 8, 17, 26, 39])
@pytest.mark.parametrize("btype", ["interaction", "each", "all"])
def test_pair_schedules_of_the_bilinear_kernels(F, btype):
    """The two host tables behind csrc/pairwise.hip / bilinear_wide.hip (interaction.py:140-156: pairs in
    itertools.combinations order): the round-robin tournament -- every round a perfect matching, every pair once, k the
    reference's pair index -- and its re-deal in groups of eight field-disjoint pairs for the fused backward (one pair per
    wave, idle entries only as padding, the tournament's k / weight index carried over)."""
    import itertools
    from deepctr_torch._hip.ops import disjoint_groups, tournament_schedule
    rows, slots, pair_w, n_w = tournament_schedule(F, btype)
    pairs = list(itertools.combinations(range(F), 2))
    P = len(pairs)
    live = [r for r in rows if r[0] >= 0]
    assert sorted(r[3] for r in live) == list(range(P)) and len(pair_w) == P
    for (i, j, w, k) in live:
        assert pairs[k] == (i, j)
        assert w == (0 if btype == "all" else (i if btype == "each" else k)) == pair_w[k]
    assert n_w == (1 if btype == "all" else (F if btype == "each" else P))
    for r0 in range(0, len(rows), slots):                    # a round: no field twice
        fields = [f for r in rows[r0:r0 + slots] if r[0] >= 0 for f in r[:2]]
        assert len(fields) == len(set(fields))
    groups = disjoint_groups(rows)
    assert all(len(g) == 8 for g in groups)
    seen = []
    for g in groups:
        fields = [f for r in g if r[0] >= 0 for f in r[:2]]
        assert len(fields) == len(set(fields))               # field-disjoint: the waves' LDS read-modify-writes never meet
        # the two halves of a group (waves 0-3 / 4-7 run half a group apart) are disjoint among themselves a fortiori
        seen += [tuple(r) for r in g if r[0] >= 0]
    assert sorted(seen) == sorted(tuple(r) for r in live)    # every pair exactly once, entries unchanged
    width = min(8, F // 2)                                   # (F fields hold at most F // 2 disjoint pairs)
    assert len(groups) <= -(-P // width) + 2                 # greedy stays near the bound
    if F == 26:
        assert len(groups) == 41
