"""Deterministic inputs and parameters of the FULL-SIZE parity fixtures (BASELINE.json configs 2-4: 26 sparse fields x
1M-row vocabularies, 13 dense, embedding_dim 16, batch 4096).

Nothing of the 1.7 GB of tables and none of the dense parameters is stored: every value is a closed-form function of its
(tensor name, flat index) through a 32-bit integer hash, evaluated with numpy integer arithmetic only, so
``oracle/make_full_golden.py`` (which runs the REAL reference on them in the build container) and the GPU tests (which
cannot see the reference) construct bit-identical tensors.  Only the reference's OUTPUTS are committed
(tests/golden/full/*.npz).  Untouched table rows cannot influence anything and are left at whatever the model
constructor drew."""
import zlib

import numpy as np

F_SPARSE, N_DENSE, DIM, VOCAB, BATCH = 26, 13, 16, 1_000_000, 4096
SEED_X, SEED_P = 20240923, 777
LR_SGD, LR_ADAGRAD, ADAGRAD_SUM0 = 0.01, 0.01, 0.05

MODELS = {
    "deepfm": dict(cls="DeepFM", kwargs=dict(dnn_hidden_units=(256, 128))),
    "xdeepfm": dict(cls="xDeepFM", kwargs=dict(dnn_hidden_units=(256, 256), cin_layer_size=(128, 128), cin_split_half=True)),
    "fibinet": dict(cls="FiBiNET", kwargs=dict(dnn_hidden_units=(128, 128), bilinear_type="interaction")),
    # round 4 (verdict: no full-size fixture had hot ids or a pooled field): the same DeepFM on ids ~ Zipf(1.05) -- a few
    # ids fill most of a batch: long duplicate segments in the update, the hot-partition path -- and with one VarLen
    # history column (maxlen 8, mean-pooled, padding id 0) next to the 26 sparse fields: the general gather / scatter path
    "deepfm_zipf": dict(cls="DeepFM", kwargs=dict(dnn_hidden_units=(256, 128)), data="zipf"),
    "deepfm_varlen": dict(cls="DeepFM", kwargs=dict(dnn_hidden_units=(256, 128)), data="varlen"),
}
HIST, HIST_LEN, HIST_COMBINER = "H", 8, "mean"


def data_of(name):
    return MODELS[name].get("data", "uniform")


def _fmix(x):
    """murmur3's 32-bit finaliser on uint64 arrays holding 32-bit values"""
    x = x & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def hash_u32(idx, stream, seed):
    """idx: integer array (< 2^32); stream, seed: ints -> uint32-valued uint64 array, well mixed"""
    idx = np.asarray(idx).astype(np.uint64)
    h = _fmix(idx ^ np.uint64((seed * 0x9E3779B1) & 0xFFFFFFFF))
    h = _fmix(h ^ np.uint64((stream * 0x7F4A7C15 + 0x165667B1) & 0xFFFFFFFF))
    return h


def unit(idx, stream, seed):
    """float32 in [0, 1) with 24 random bits: exactly representable"""
    return (hash_u32(idx, stream, seed) >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0)


def sym(idx, stream, seed, scale):
    """float32 uniform in (-scale/2, scale/2): two exactly rounded fp32 operations"""
    return (unit(idx, stream, seed) - np.float32(0.5)) * np.float32(scale)


def stream_of(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def n_xcols(data="uniform"):
    return F_SPARSE + N_DENSE + (HIST_LEN if data == "varlen" else 0)


def inputs(data="uniform"):
    """X [B, 26 (+ 8) + 13] float32 (ids as floats, exact below 2^24; dense in [0, 1)), y [B] float32.
    data: "uniform" ids (+ a planted block of duplicates) | "zipf" ids ~ Zipf(1.05) | "varlen": uniform ids + 8 history
    columns between the sparse and the dense ones."""
    b = np.arange(BATCH, dtype=np.int64)
    X = np.zeros((BATCH, n_xcols(data)), np.float32)
    for f in range(F_SPARSE):
        if data == "zipf":
            a = 1.05            # inverse CDF of the continuous approximation (bench.py --ids zipf), in fp64
            u = (hash_u32(b, 1000 + f, SEED_X) >> np.uint64(8)).astype(np.float64) / 16777216.0
            ids = np.clip(np.floor(((float(VOCAB) ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))), 1, VOCAB).astype(np.int64) - 1
        else:
            ids = (hash_u32(b, 1000 + f, SEED_X) % np.uint64(VOCAB)).astype(np.int64)
            ids[1:BATCH // 16] = np.where(np.arange(1, BATCH // 16) % (f + 2) == 0, ids[0], ids[1:BATCH // 16])  # duplicates
        X[:, f] = ids.astype(np.float32)
    d0 = F_SPARSE
    if data == "varlen":
        length = (hash_u32(b, 4000, SEED_X) % np.uint64(HIST_LEN + 1)).astype(np.int64)       # 0 .. 8 valid positions
        for t in range(HIST_LEN):
            ids = 1 + (hash_u32(b, 4100 + t, SEED_X) % np.uint64(VOCAB - 1)).astype(np.int64)
            X[:, F_SPARSE + t] = np.where(t < length, ids, 0).astype(np.float32)               # 0 = padding (inputs.py:146)
        d0 += HIST_LEN
    for j in range(N_DENSE):
        X[:, d0 + j] = unit(b, 2000 + j, SEED_X)
    y = (hash_u32(b, 3000, SEED_X) & np.uint64(1)).astype(np.float32)
    return X, y


def param_scale(name, shape):
    """std 0.05 for every table ('trained-like': |logit| of O(1)), fan-in scaled dense weights, small biases"""
    if "embedding_dict" in name:
        return 0.05 * np.sqrt(12.0)
    if name.endswith("bias"):
        return 0.05 * np.sqrt(12.0)
    if name == "linear_model.weight":
        return 0.3 * np.sqrt(12.0)
    fan_in = shape[1] if len(shape) >= 2 else shape[0]
    return (1.2 / np.sqrt(max(1, fan_in))) * np.sqrt(12.0)


def dense_param(name, shape):
    n = int(np.prod(shape))
    return sym(np.arange(n, dtype=np.int64), stream_of(name), SEED_P, param_scale(name, shape)).reshape(shape)


def table_rows(name, rows, dim):
    """values of the given rows of table `name` ([len(rows), dim]): element (r, d) is flat index r * dim + d"""
    rows = np.asarray(rows, np.int64)
    idx = rows[:, None] * dim + np.arange(dim, dtype=np.int64)[None, :]
    return sym(idx, stream_of(name), SEED_P, param_scale(name, (VOCAB, dim)))


def touched_rows(X, data="uniform"):
    """per table (the 26 sparse fields, then the history table): the sorted unique ids of the batch"""
    t = [np.unique(X[:, f].astype(np.int64)) for f in range(F_SPARSE)]
    if data == "varlen":
        t.append(np.unique(X[:, F_SPARSE:F_SPARSE + HIST_LEN].astype(np.int64)))    # (row 0: read, masked, gradient 0)
    return t


def column_names():
    return ["C%d" % (i + 1) for i in range(F_SPARSE)], ["I%d" % (i + 1) for i in range(N_DENSE)]


def table_names(data="uniform"):
    return column_names()[0] + ([HIST] if data == "varlen" else [])


def feature_columns(mod, data="uniform"):
    """the model's feature columns from the SparseFeat / VarLenSparseFeat / DenseFeat of `mod` (the reference's
    deepctr_torch.inputs or the drop-in's), in X's column order"""
    sparse, dense = column_names()
    cols = [mod.SparseFeat(c, VOCAB, DIM) for c in sparse]
    if data == "varlen":
        cols.append(mod.VarLenSparseFeat(mod.SparseFeat(HIST, VOCAB, DIM), maxlen=HIST_LEN, combiner=HIST_COMBINER))
    return cols + [mod.DenseFeat(c, 1) for c in dense]


# ---- what of a big tensor is stored / compared -----------------------------------------------------------------------
BIG = 300_000          # tensors above this many elements: a strided sample + 4 random projections instead of all values
BIG_STEP = 20_000      # ... for the parameters after a train step (every gradient element is already pinned in full)
STRIDE = 8
ROW_KEEP = 16          # touched table rows with id % ROW_KEEP == 0 are stored in full; all rows enter the projections


def proj_weights(n, k):
    return sym(np.arange(n, dtype=np.int64), 9000 + k, SEED_P, 2.0).astype(np.float64)


def summarise(arr, big=BIG):
    """{'all': values} for small tensors, {'sample': strided values, 'proj': 4 dot products in fp64} for big ones"""
    a = np.asarray(arr)
    flat = a.reshape(-1)
    if flat.size <= big:
        return {"all": a.astype(np.float32)}
    return {"sample": flat[::STRIDE].astype(np.float32),
            "proj": np.array([float(np.dot(flat.astype(np.float64), proj_weights(flat.size, k))) for k in range(4)]),
            "absmax": np.array(float(np.abs(flat).max()))}


def summarise_rows(rows, values):
    """touched rows of a deep table: the rows with id % ROW_KEEP == 0 in full + 4 projections over all touched rows"""
    rows = np.asarray(rows, np.int64)
    v = np.asarray(values, np.float64)
    keep = rows % ROW_KEEP == 0
    idx = rows[:, None] * v.shape[1] + np.arange(v.shape[1], dtype=np.int64)[None, :]
    proj = np.array([float(np.sum(v * sym(idx, 9100 + k, SEED_P, 2.0).astype(np.float64))) for k in range(4)])
    return {"rows": rows[keep], "values": v[keep].astype(np.float32), "proj": proj, "absmax": np.array(float(np.abs(v).max()))}
