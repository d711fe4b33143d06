# -*- coding: utf-8 -*-
"""Criteo-format text -> binary shards -> model input (the data format on the caller's side of the
hot path).

The reference's example (``examples/run_classification_criteo.py:12-26``) reads the whole CSV with pandas, fills
missing sparse values with ``'-1'`` and missing dense values with 0, label-encodes every sparse column
(``sklearn.preprocessing.LabelEncoder``: classes = sorted unique values) and min-max scales every dense column
(``MinMaxScaler(feature_range=(0, 1))``), then hands ``model.fit`` a dict of columns.  That is per-epoch-free but
single-shot and in-memory; the 45 M-row Criteo files do not fit that way.  Here the same transformation is a
streaming two-pass encoder that writes fixed-width binary shards, and a reader that maps them back to exactly the
arrays the example would have produced:

    meta = encode_criteo("train.txt", "shards/", rows_per_shard=1 << 22)      # pass 1: vocabularies + min/max; pass 2: write
    ds = CriteoShards("shards/")
    cols = ds.feature_columns(embedding_dim=16)                               # SparseFeat / DenseFeat like the example
    model.fit(ds.model_input(), ds.labels(), batch_size=4096)                 # dict name -> array, ids as int32

Shard layout (little endian): 32-byte header ``b"DCTRSHD1", n_rows u64, n_sparse u32, n_dense u32, reserved u64``, then
``ids int32 [n_rows, n_sparse]``, ``dense float32 [n_rows, n_dense]``, ``label float32 [n_rows]`` -- three
contiguous blocks, so a shard maps straight into the ``X [N, 39]`` float matrix the gather kernel reads (ids are
converted to float32 on upload, exactly what ``BaseModel.fit`` does with the example's columns, basemodel.py:155-156).
"""
import json
import os
import struct

import numpy as np

MAGIC = b"DCTRSHD1"
_HEADER = struct.Struct("<8sQIIQ")


def _open_text(path):
    import io
    if hasattr(path, "read"):
        return path
    return io.open(path, "r", encoding="utf-8", newline="")


def _rows(path, sep, has_header):
    """Yield lists of raw string fields; the header (if any) is skipped."""
    f = _open_text(path)
    try:
        first = True
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if first and has_header:
                first = False
                continue
            first = False
            yield line.split(sep)
    finally:
        if not hasattr(path, "read"):
            f.close()


def sniff(path):
    """(separator, has_header) of a Criteo-format text file: the Kaggle / Terabyte dumps are tab separated without a
    header, the reference's ``criteo_sample.txt`` is comma separated with one."""
    f = _open_text(path)
    try:
        line = f.readline()
    finally:
        if not hasattr(path, "read"):
            f.close()
        else:
            f.seek(0)
    sep = "\t" if line.count("\t") >= line.count(",") else ","
    return sep, line.lower().startswith("label")


def encode_criteo(path, out_dir, n_dense=13, n_sparse=26, rows_per_shard=1 << 22, sep=None, has_header=None,
                  sparse_fill="-1"):
    """Two streaming passes over a Criteo-format file ``label, I1..I{n_dense}, C1..C{n_sparse}``.

    Pass 1 collects, per sparse column, the set of values (missing -> ``sparse_fill``, like ``fillna('-1')``) and, per
    dense column, min / max (missing -> 0, like ``fillna(0)``).  Pass 2 writes the shards: id = rank of the value in
    the SORTED vocabulary (``LabelEncoder``), dense = (x - min) / (max - min) with a zero range mapped like
    ``MinMaxScaler`` does (scale 1).  Returns the meta dict (also written to ``out_dir/meta.json``)."""
    if sep is None or has_header is None:
        s, h = sniff(path)
        sep = s if sep is None else sep
        has_header = h if has_header is None else has_header
    os.makedirs(out_dir, exist_ok=True)
    vocab = [set() for _ in range(n_sparse)]
    lo = np.full(n_dense, np.inf)
    hi = np.full(n_dense, -np.inf)
    n_rows = 0
    for fields in _rows(path, sep, has_header):
        if len(fields) != 1 + n_dense + n_sparse:
            raise ValueError("row %d has %d fields, expected %d" % (n_rows, len(fields), 1 + n_dense + n_sparse))
        for j in range(n_dense):
            v = fields[1 + j]
            x = float(v) if v != "" else 0.0
            if x < lo[j]:
                lo[j] = x
            if x > hi[j]:
                hi[j] = x
        for j in range(n_sparse):
            v = fields[1 + n_dense + j]
            vocab[j].add(v if v != "" else sparse_fill)
        n_rows += 1
    if n_rows == 0:
        raise ValueError("empty input")
    classes = [sorted(v) for v in vocab]                 # LabelEncoder: np.unique -> sorted
    index = [dict((c, i) for i, c in enumerate(cl)) for cl in classes]
    rng = hi - lo
    scale = np.where(rng == 0, 1.0, rng)                 # MinMaxScaler's _handle_zeros_in_scale

    shards, buf_ids, buf_dense, buf_y = [], [], [], []

    def flush():
        if not buf_y:
            return
        ids = np.asarray(buf_ids, dtype=np.int32).reshape(-1, n_sparse)
        dense = np.asarray(buf_dense, dtype=np.float64).reshape(-1, n_dense)
        dense = ((dense - lo) / scale).astype(np.float32)            # scaled in float64 like sklearn, stored as fp32
        y = np.asarray(buf_y, dtype=np.float32)
        name = "shard_%05d.bin" % len(shards)
        with open(os.path.join(out_dir, name), "wb") as fh:
            fh.write(_HEADER.pack(MAGIC, ids.shape[0], n_sparse, n_dense, 0))
            fh.write(ids.tobytes())
            fh.write(dense.tobytes())
            fh.write(y.tobytes())
        shards.append({"file": name, "rows": int(ids.shape[0])})
        del buf_ids[:], buf_dense[:], buf_y[:]

    for fields in _rows(path, sep, has_header):
        buf_y.append(float(fields[0]))
        buf_dense.append([float(v) if v != "" else 0.0 for v in fields[1:1 + n_dense]])
        buf_ids.append([index[j][fields[1 + n_dense + j] if fields[1 + n_dense + j] != "" else sparse_fill]
                        for j in range(n_sparse)])
        if len(buf_y) >= rows_per_shard:
            flush()
    flush()
    meta = {"format": MAGIC.decode(), "rows": n_rows, "n_dense": n_dense, "n_sparse": n_sparse,
            "dense_names": ["I%d" % (i + 1) for i in range(n_dense)],
            "sparse_names": ["C%d" % (i + 1) for i in range(n_sparse)],
            "vocabulary_sizes": [len(c) for c in classes], "dense_min": lo.tolist(), "dense_max": hi.tolist(),
            "shards": shards}
    with open(os.path.join(out_dir, "meta.json"), "w") as fh:
        json.dump(meta, fh)
    for j, cl in enumerate(classes):                     # one value per line, rank = line number (decoding / serving)
        with open(os.path.join(out_dir, "vocab_C%d.txt" % (j + 1)), "w", encoding="utf-8") as fh:
            fh.write("\n".join(cl))
    return meta


def read_shard(path):
    """(ids int32 [n, n_sparse], dense float32 [n, n_dense], label float32 [n]) as read-only memory maps."""
    with open(path, "rb") as fh:
        magic, n, ns, nd, _ = _HEADER.unpack(fh.read(_HEADER.size))
    if magic != MAGIC:
        raise ValueError("%s is not a %s shard" % (path, MAGIC.decode()))
    off = _HEADER.size
    ids = np.memmap(path, dtype=np.int32, mode="r", offset=off, shape=(n, ns))
    off += 4 * n * ns
    dense = np.memmap(path, dtype=np.float32, mode="r", offset=off, shape=(n, nd))
    off += 4 * n * nd
    y = np.memmap(path, dtype=np.float32, mode="r", offset=off, shape=(n,))
    return ids, dense, y


class CriteoShards(object):
    """A directory written by ``encode_criteo``: feature columns, model input, labels, and the device-resident
    ``X [N, n_sparse + n_dense]`` float32 matrix ``BaseModel.fit`` builds from the model input anyway."""

    def __init__(self, directory):
        self.directory = directory
        with open(os.path.join(directory, "meta.json")) as fh:
            self.meta = json.load(fh)
        if self.meta.get("format") != MAGIC.decode():
            raise ValueError("unknown shard format %r" % (self.meta.get("format"),))
        self._parts = None

    def __len__(self):
        return int(self.meta["rows"])

    def parts(self):
        if self._parts is None:
            self._parts = [read_shard(os.path.join(self.directory, s["file"])) for s in self.meta["shards"]]
        return self._parts

    def feature_columns(self, embedding_dim=4):
        """``[SparseFeat(C1, vocab, dim), ..., DenseFeat(I1, 1), ...]`` as the reference's example builds them
        (run_classification_criteo.py:30-32)."""
        from .inputs import DenseFeat, SparseFeat
        m = self.meta
        return [SparseFeat(n, vocabulary_size=v, embedding_dim=embedding_dim)
                for n, v in zip(m["sparse_names"], m["vocabulary_sizes"])] + \
               [DenseFeat(n, 1) for n in m["dense_names"]]

    def model_input(self):
        """dict feature name -> 1-D array over all shards (the ``train_model_input`` of the example)."""
        m = self.meta
        ids = np.concatenate([p[0] for p in self.parts()], axis=0)
        dense = np.concatenate([p[1] for p in self.parts()], axis=0)
        out = {n: ids[:, j] for j, n in enumerate(m["sparse_names"])}
        out.update({n: dense[:, j] for j, n in enumerate(m["dense_names"])})
        return out

    def labels(self):
        return np.concatenate([p[2] for p in self.parts()], axis=0)

    def matrix(self, device, feature_index=None):
        """``X [N, W]`` float32 on ``device`` in ``feature_index`` order (default: sparse columns then dense ones, the
        order ``build_input_features`` gives the example's column list) and ``y [N]``: what ``fit`` keeps resident."""
        import torch
        m = self.meta
        names = m["sparse_names"] + m["dense_names"]
        if feature_index is not None:
            names = sorted(names, key=lambda n: feature_index[n][0])
        cols = self.model_input()
        X = torch.empty((len(self), len(names)), dtype=torch.float32, device=device)
        for j, n in enumerate(names):
            X[:, j] = torch.from_numpy(np.ascontiguousarray(cols[n])).to(device).float()
        return X, torch.from_numpy(np.ascontiguousarray(self.labels())).to(device)


__all__ = ["encode_criteo", "read_shard", "CriteoShards", "sniff"]
