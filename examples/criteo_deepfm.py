"""DeepFM on a Criteo-format file with the drop-in package on an MI355X -- the workflow of the reference's
``examples/run_classification_criteo.py`` (label-encode the 26 categorical columns, min-max scale the 13 numeric ones,
80/20 split, DeepFM, adagrad, LogLoss / AUC on the held-out rows), with the preprocessing done by the streaming shard
encoder so that it also works for files that do not fit a pandas frame.

    python examples/criteo_deepfm.py /path/to/criteo_sample.txt --epochs 10 --batch-size 32
    python examples/criteo_deepfm.py /data/criteo/train.txt --shards /data/criteo/shards --embedding-dim 16 --batch-size 4096
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepctr-torch_amd"))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("text", help="label, I1..I13, C1..C26 per row (comma separated with a header, or the tab separated dumps)")
    ap.add_argument("--shards", default=None, help="directory for the binary shards (default: a temporary one)")
    ap.add_argument("--embedding-dim", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--optimizer", default="adagrad")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()

    from sklearn.metrics import log_loss, roc_auc_score
    from sklearn.model_selection import train_test_split
    from deepctr_torch.data import CriteoShards, encode_criteo
    from deepctr_torch.inputs import get_feature_names
    from deepctr_torch.models import DeepFM

    out = args.shards or tempfile.mkdtemp(prefix="criteo_shards_")
    if not os.path.exists(os.path.join(out, "meta.json")):
        meta = encode_criteo(args.text, out)            # two streaming passes: vocabularies / ranges, then the shards
        print("encoded %d rows into %d shard(s) under %s" % (meta["rows"], len(meta["shards"]), out))
    ds = CriteoShards(out)
    columns = ds.feature_columns(embedding_dim=args.embedding_dim)
    names = get_feature_names(columns)
    x, y = ds.model_input(), ds.labels()

    idx_train, idx_test = train_test_split(np.arange(len(ds)), test_size=0.2, random_state=2020)
    train = {n: np.asarray(x[n])[idx_train] for n in names}
    test = {n: np.asarray(x[n])[idx_test] for n in names}

    model = DeepFM(linear_feature_columns=columns, dnn_feature_columns=columns, task="binary", l2_reg_embedding=1e-5,
                   device=args.device)
    model.compile(args.optimizer, "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    model.fit(train, y[idx_train], batch_size=args.batch_size, epochs=args.epochs, verbose=2, validation_split=0.2)
    pred = model.predict(test, 256)
    print("test LogLoss", round(log_loss(y[idx_test], pred), 4))
    print("test AUC", round(roc_auc_score(y[idx_test], pred), 4))


if __name__ == "__main__":
    main()
