"""BASELINE.json configs[0]: the reference's own Criteo example (examples/run_classification_criteo.py on
examples/criteo_sample.txt, 200 rows) executed with the REAL reference on torch-CPU, step for step as the example does it:
pandas read_csv, fillna, LabelEncoder per sparse column, MinMaxScaler over the dense ones, SparseFeat(embedding_dim=4) /
DenseFeat columns, train_test_split(test_size=0.2, random_state=2020), DeepFM(l2_reg_embedding=1e-5),
compile('adagrad', 'binary_crossentropy', ['binary_crossentropy', 'auc']), fit(batch_size=32, epochs=10,
validation_split=0.2) with the default shuffle=True, predict(test, 256), test LogLoss / AUC.

One line is added: torch.manual_seed(FIT_SEED) right before fit(), so that the shuffles do not depend on how many random
numbers model construction consumed.  Stored in tests/golden/api/criteo_example.npz: the encoded train / test matrices
(ids and scaled values -- derived data, not the text), the freshly initialised state_dict, History, predictions, test
metrics.  Also checked here: ``deepctr_torch.data.encode_criteo`` (this repo's streaming encoder) produces exactly the
example's ids and scaled values from the same text.

    python oracle/check_criteo_example.py        # build container only (needs /root/reference)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

FIT_SEED = 2020
SAMPLE = os.path.join(mg.REFERENCE, "examples", "criteo_sample.txt")


def encode_with_this_repo(tmp):
    """Run the drop-in's encoder in a subprocess (both packages are called deepctr_torch) -> ids, dense, label."""
    code = ("import sys; sys.path.insert(0, %r); from deepctr_torch.data import encode_criteo; "
            "encode_criteo(%r, %r)" % (os.path.join(ROOT, "deepctr-torch_amd"), SAMPLE, tmp))
    subprocess.check_call([sys.executable, "-c", code])
    meta = json.load(open(os.path.join(tmp, "meta.json")))
    parts = []
    for s in meta["shards"]:
        raw = open(os.path.join(tmp, s["file"]), "rb").read()
        n, ns, nd = s["rows"], meta["n_sparse"], meta["n_dense"]
        off = 32
        ids = np.frombuffer(raw, np.int32, n * ns, off).reshape(n, ns)
        off += 4 * n * ns
        dense = np.frombuffer(raw, np.float32, n * nd, off).reshape(n, nd)
        off += 4 * n * nd
        parts.append((ids, dense, np.frombuffer(raw, np.float32, n, off)))
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]), meta)


def main():
    import pandas as pd
    import torch
    from sklearn.metrics import log_loss, roc_auc_score
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import LabelEncoder, MinMaxScaler
    with tempfile.TemporaryDirectory() as tmp:
        my_ids, my_dense, my_label, my_meta = encode_with_this_repo(tmp)

    mg.import_reference()
    from deepctr_torch.inputs import DenseFeat, SparseFeat, get_feature_names
    from deepctr_torch.models import DeepFM
    data = pd.read_csv(SAMPLE)
    sparse_features = ['C' + str(i) for i in range(1, 27)]
    dense_features = ['I' + str(i) for i in range(1, 14)]
    data[sparse_features] = data[sparse_features].fillna('-1')
    data[dense_features] = data[dense_features].fillna(0)
    for feat in sparse_features:
        data[feat] = LabelEncoder().fit_transform(data[feat])
    data[dense_features] = MinMaxScaler(feature_range=(0, 1)).fit_transform(data[dense_features])

    # this repo's encoder == the example's preprocessing, value for value
    assert np.array_equal(my_ids, data[sparse_features].values.astype(np.int32))
    assert np.array_equal(my_dense, data[dense_features].values.astype(np.float32))
    assert np.array_equal(my_label, data['label'].values.astype(np.float32))
    assert my_meta["vocabulary_sizes"] == [int(data[f].max()) + 1 for f in sparse_features]
    print("encode_criteo == LabelEncoder + MinMaxScaler on criteo_sample.txt (%d rows)" % len(data))

    cols = [SparseFeat(f, vocabulary_size=data[f].max() + 1, embedding_dim=4) for f in sparse_features] + \
           [DenseFeat(f, 1) for f in dense_features]
    names = get_feature_names(cols + cols)
    train, test = train_test_split(data, test_size=0.2, random_state=2020)
    model = DeepFM(linear_feature_columns=cols, dnn_feature_columns=cols, task='binary', l2_reg_embedding=1e-5,
                   device='cpu')
    init = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    model.compile("adagrad", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    torch.manual_seed(FIT_SEED)
    hist = model.fit({n: train[n] for n in names}, train[['label']].values, batch_size=32, epochs=10, verbose=2,
                     validation_split=0.2)
    pred = model.predict({n: test[n] for n in names}, 256)
    ll, auc = log_loss(test[['label']].values, pred), roc_auc_score(test[['label']].values, pred)
    print("test LogLoss", round(ll, 4), "test AUC", round(auc, 4))

    store = {"names": np.array(json.dumps(names)),
             "vocab": np.asarray([int(data[f].max()) + 1 for f in sparse_features], np.int64),
             "train_X": train[names].values.astype(np.float64), "train_y": train[['label']].values.astype(np.float64),
             "test_X": test[names].values.astype(np.float64), "test_y": test[['label']].values.astype(np.float64),
             "pred": pred, "test_logloss": np.array(ll), "test_auc": np.array(auc)}
    for k, v in init.items():
        store["param/" + k] = v
    for k, v in hist.history.items():
        store["hist/" + k] = np.asarray(v, np.float64)
    out = os.path.join(ROOT, "tests", "golden", "api")
    os.makedirs(out, exist_ok=True)
    np.savez_compressed(os.path.join(out, "criteo_example.npz"), **store)
    print({k: np.round(v, 4).tolist() for k, v in hist.history.items()})


if __name__ == "__main__":
    main()
