"""The data generator of the reference's model tests, restated (tests/utils.py:19-66 ``get_test_data``), and the helpers
around it.  Shared by tests/test_gpu_reference_matrix.py and oracle/check_matrix.py (which runs the real reference on the
same draws and stores its answers)."""
import numpy as np
import torch

N = 64


def make_data(seed, n_sparse, n_dense, emb=4, seqs=("sum", "mean", "max"), include_length=False, min_clean=16):
    """The reference's generator restated; re-drawn (seed + 1000, ...) until at least ``min_clean`` rows have a well
    defined forward value (see ``clean_rows``), so that every configuration gets a numeric check."""
    while True:
        x, y, cols = _draw(seed, n_sparse, n_dense, emb, seqs, include_length)
        if int(clean_rows(x, cols).sum()) >= min_clean:
            return x, y, cols
        seed += 1000


def _draw(seed, n_sparse, n_dense, emb, seqs, include_length):
    from deepctr_torch.inputs import DenseFeat, SparseFeat, VarLenSparseFeat
    rng = np.random.default_rng(seed)
    cols, x = [], {}
    for i in range(n_sparse):
        cols.append(SparseFeat("sparse_feature_%d" % i, int(rng.integers(1, 10)), emb, dtype=torch.int32))
    for i in range(n_dense):
        cols.append(DenseFeat("dense_feature_%d" % i, 1, dtype=torch.float32))
    for mode in seqs:
        ln = "sequence_%s_seq_length" % mode if include_length else None
        cols.append(VarLenSparseFeat(SparseFeat("sequence_" + mode, vocabulary_size=int(rng.integers(1, 10)),
                                                embedding_dim=emb), maxlen=int(rng.integers(1, 10)), combiner=mode,
                                     length_name=ln))
    for fc in cols:
        if isinstance(fc, VarLenSparseFeat):
            x[fc.name] = rng.integers(0, fc.vocabulary_size, (N, fc.maxlen))
            if fc.length_name:
                x[fc.length_name] = rng.integers(1, fc.maxlen + 1, N)
        elif isinstance(fc, SparseFeat):
            x[fc.name] = rng.integers(0, fc.vocabulary_size, N)
        else:
            x[fc.name] = rng.random(N)
    return x, rng.integers(0, 2, N), cols


def clean_rows(x, cols):
    """Rows whose forward value is well defined.  A 'max' VarLen field whose ids are ALL the padding id 0 pools to
    ``embedding - 1e9`` in the reference (sequence.py:65-68 subtracts 1e9 from masked positions and takes the max): the
    logit is then ~1e9 (FM: ~1e18) and its low digits are fp32 rounding noise in the reference itself.  The generator
    of the reference's tests produces such rows (vocabulary 1, or short sequences of zeros); they must run -- and do,
    through fit / predict below -- but a numeric comparison is only meaningful on the others."""
    ok = np.ones(N, bool)
    for c in cols:
        if getattr(c, "combiner", None) == "max" and not c.length_name:
            ok &= (np.asarray(x[c.name]) != 0).any(axis=1)
    return ok


def spec_of(model_name, lin, dnn, **kwargs):
    def one(c):
        if hasattr(c, "dimension"):
            return {"kind": "dense", "name": c.name, "dimension": c.dimension}
        d = {"kind": "sparse", "name": c.name, "vocab": c.vocabulary_size, "dim": c.embedding_dim,
             "embedding_name": c.embedding_name}
        if hasattr(c, "maxlen"):
            d.update(kind="varlen", maxlen=c.maxlen, combiner=c.combiner, length_name=c.length_name)
        return d
    return {"model": model_name, "linear_columns": [one(c) for c in lin], "dnn_columns": [one(c) for c in dnn],
            "kwargs": kwargs}
