"""CPU: the Criteo text -> binary shard encoder (deepctr_torch/data.py) against the reference example's own recipe
(examples/run_classification_criteo.py:12-26: pandas fillna + sklearn LabelEncoder + MinMaxScaler)."""
import io
import os

import numpy as np
import pandas as pd
import pytest
from sklearn.preprocessing import LabelEncoder, MinMaxScaler

from helpers import GOLDEN_DIR  # noqa: F401  (path setup via conftest)


def _synthetic_csv(n=257, seed=0, sep=",", header=True):
    rng = np.random.default_rng(seed)
    dense = ["I%d" % i for i in range(1, 14)]
    sparse = ["C%d" % i for i in range(1, 27)]
    rows = []
    for _ in range(n):
        r = [str(int(rng.integers(0, 2)))]
        for j in range(13):
            r.append("" if rng.random() < 0.2 else ("%d" % rng.integers(-3, 500) if j % 2 else "%.1f" % (rng.random() * 1e4)))
        for j in range(26):
            r.append("" if rng.random() < 0.15 else "%08x" % rng.integers(0, 40 + 13 * j))
        rows.append(sep.join(r))
    text = (sep.join(["label"] + dense + sparse) + "\n" if header else "") + "\n".join(rows) + "\n"
    return text, dense, sparse


def _reference_recipe(text, dense, sparse):
    data = pd.read_csv(io.StringIO(text), dtype={c: str for c in sparse})
    data[sparse] = data[sparse].fillna("-1")
    data[dense] = data[dense].fillna(0)
    for feat in sparse:
        data[feat] = LabelEncoder().fit_transform(data[feat])
    data[dense] = MinMaxScaler(feature_range=(0, 1)).fit_transform(data[dense])
    return data


@pytest.mark.parametrize("rows_per_shard", [64, 1 << 20])
def test_encoder_matches_the_examples_preprocessing(tmp_path, rows_per_shard):
    from deepctr_torch.data import CriteoShards, encode_criteo
    text, dense, sparse = _synthetic_csv()
    src = tmp_path / "criteo.csv"
    src.write_text(text)
    meta = encode_criteo(str(src), str(tmp_path / "shards"), rows_per_shard=rows_per_shard)
    ref = _reference_recipe(text, dense, sparse)
    ds = CriteoShards(str(tmp_path / "shards"))
    assert len(ds) == len(ref) == meta["rows"]
    assert len(meta["shards"]) == (5 if rows_per_shard == 64 else 1)
    cols = ds.model_input()
    for c in sparse:
        np.testing.assert_array_equal(cols[c], ref[c].to_numpy().astype(np.int32), err_msg=c)
    for c in dense:
        np.testing.assert_allclose(cols[c], ref[c].to_numpy().astype(np.float32), rtol=0, atol=1e-7, err_msg=c)
    np.testing.assert_array_equal(ds.labels(), ref["label"].to_numpy().astype(np.float32))
    # the feature columns the example derives: vocabulary_size = max id + 1
    fcs = ds.feature_columns(embedding_dim=4)
    assert [fc.vocabulary_size for fc in fcs[:26]] == [int(ref[c].max()) + 1 for c in sparse]
    assert [fc.name for fc in fcs] == sparse + dense


def test_tab_separated_headerless_input_and_matrix(tmp_path):
    """The Kaggle / Terabyte dumps: tab separated, no header.  matrix() gives X in feature_index order."""
    import torch
    from deepctr_torch.data import CriteoShards, encode_criteo, sniff
    from deepctr_torch.inputs import build_input_features
    text, dense, sparse = _synthetic_csv(n=50, seed=3, sep="\t", header=False)
    src = tmp_path / "day_0.tsv"
    src.write_text(text)
    assert sniff(str(src)) == ("\t", False)
    encode_criteo(str(src), str(tmp_path / "s"))
    ds = CriteoShards(str(tmp_path / "s"))
    fi = build_input_features(ds.feature_columns())
    X, y = ds.matrix("cpu", fi)
    assert X.shape == (50, 39) and X.dtype == torch.float32 and y.shape == (50,)
    cols = ds.model_input()
    np.testing.assert_array_equal(X[:, fi["C3"][0]].numpy(), cols["C3"].astype(np.float32))
    np.testing.assert_array_equal(X[:, fi["I5"][0]].numpy(), cols["I5"])
    assert float(X[:, 26:].min()) >= 0.0 and float(X[:, 26:].max()) <= 1.0


def test_bad_inputs_raise(tmp_path):
    from deepctr_torch.data import encode_criteo, read_shard
    p = tmp_path / "bad.csv"
    p.write_text("label,I1\n1,2,3\n")
    with pytest.raises(ValueError):
        encode_criteo(str(p), str(tmp_path / "o"))
    q = tmp_path / "junk.bin"
    q.write_bytes(b"\0" * 64)
    with pytest.raises(ValueError):
        read_shard(str(q))
