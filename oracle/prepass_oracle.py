"""numpy restatement of the id-only pre-pass of the deterministic embedding update (test infrastructure; only tests/ may
import it).

What the update consumes (deepctr-torch_amd/csrc/update.hip, header comment and k_embed_segments): for every unit (one id
column) the batch's entries are dealt to P partitions by ``id mod P``; a partition's bucket holds the 32-bit keys
``(id // P) << bbits | sample`` SORTED ascending (= by id, then by sample), with the count beside it; a partition of more
than ``bucket`` entries only gets its count (the update kernel's general path takes it).  That is the contract
``segments_direct`` states.  ``segments_two_level`` follows the large-batch kernels step by step (k_prepass_bin: a chunk of
samples counting-sorted by coarse bin of ``fine`` consecutive partitions, keys + partition tags written back in bin order
with the chunk's bin offsets; k_prepass_sort: per bin, the runs of all chunks dealt to ``fine`` buckets and rank-sorted) --
the two must agree for ANY ids, chunk size and bin width.  The reference itself has no counterpart: it materialises dense
[V, D] gradients through aten::embedding_dense_backward (basemodel.py:261-262)."""
import numpy as np


def ceil_log2(x):
    l = 0
    while (1 << l) < x:
        l += 1
    return l


def segments_direct(ids, vocab, P, bucket=512):
    """ids [B] int -> (counts [P], keys: list of P sorted uint32 arrays, empty for an overflowing partition)"""
    ids = np.asarray(ids, dtype=np.int64)
    B = ids.shape[0]
    bbits = ceil_log2(max(B, 2))
    ids = np.where((ids < 0) | (ids >= vocab), 0, ids)            # clamp_id: out-of-range ids read row 0
    part = ids % P
    key = ((ids // P) << bbits) | np.arange(B, dtype=np.int64)
    counts = np.bincount(part, minlength=P).astype(np.int64)
    keys = []
    for p in range(P):
        k = np.sort(key[part == p]).astype(np.uint32)
        keys.append(k if counts[p] <= bucket else np.zeros(0, dtype=np.uint32))
    return counts, keys


def segments_two_level(ids, vocab, P, chunk, fine, bucket=512, rng=None):
    """The same result through the two kernels' steps.  ``rng``: shuffles the order inside every (chunk, bin) run and
    inside every bucket before the rank sort -- the kernels' LDS atomics hand out slots in no particular order, the rank
    sort must not care."""
    ids = np.asarray(ids, dtype=np.int64)
    B = ids.shape[0]
    bbits = ceil_log2(max(B, 2))
    ids = np.where((ids < 0) | (ids >= vocab), 0, ids)
    n_bins = (P + fine - 1) // fine
    n_chunks = (B + chunk - 1) // chunk
    # ---- level 1 (k_prepass_bin): every chunk re-ordered by coarse bin, with its bin offsets
    stage_keys, stage_tags, offs = [], [], []
    for ck in range(n_chunks):
        lo, hi = ck * chunk, min(B, (ck + 1) * chunk)
        c_ids = ids[lo:hi]
        p = c_ids % P
        key = ((c_ids // P) << bbits) | np.arange(lo, hi, dtype=np.int64)
        b = p // fine
        cnt = np.bincount(b, minlength=n_bins)
        start = np.concatenate([[0], np.cumsum(cnt)])
        order = np.argsort(b, kind="stable")
        if rng is not None:                                   # any order inside a bin's run
            for c in range(n_bins):
                seg = order[start[c]:start[c + 1]].copy()
                rng.shuffle(seg)
                order[start[c]:start[c + 1]] = seg
        stage_keys.append(key[order])
        stage_tags.append(p[order])
        offs.append(start)
    # ---- level 2 (k_prepass_sort): per bin, the runs of all chunks into `fine` buckets, rank sort of each
    counts = np.zeros(P, dtype=np.int64)
    keys = [np.zeros(0, dtype=np.uint32) for _ in range(P)]
    for c in range(n_bins):
        buckets = [[] for _ in range(fine)]
        for ck in range(n_chunks):
            s0, s1 = offs[ck][c], offs[ck][c + 1]
            for k, t in zip(stage_keys[ck][s0:s1], stage_tags[ck][s0:s1]):
                buckets[int(t) - c * fine].append(int(k))
        for f in range(fine):
            p = c * fine + f
            if p >= P:
                assert not buckets[f]
                continue
            counts[p] = len(buckets[f])
            if len(buckets[f]) > bucket:
                continue                                      # (the kernel keeps the first `bucket` keys and never uses them)
            arr = np.asarray(buckets[f], dtype=np.int64)
            if rng is not None:
                rng.shuffle(arr)
            rank = (arr[None, :] < arr[:, None]).sum(1)       # keys are unique: rank = number of smaller keys
            out = np.zeros(len(arr), dtype=np.uint32)
            out[rank] = arr.astype(np.uint32)
            keys[p] = out
    return counts, keys
