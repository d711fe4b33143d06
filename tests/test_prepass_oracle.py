"""CPU: the numpy restatement of the update's id pre-pass (oracle/prepass_oracle.py) -- the two-level path of large batches
(csrc/update.hip k_prepass_bin + k_prepass_sort) must produce the buckets of the direct statement for any ids, chunk size
and bin width, whatever order the slots inside a run / bucket are handed out in.  The GPU side of the same statement:
tests/test_gpu_update.py::test_prepass_layouts_and_inkernel_scan_agree_bit_for_bit (kernels against the in-kernel scan)
and tools/prepass_bench.py (old path against new, bit for bit, up to B = 262 144)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from prepass_oracle import segments_direct, segments_two_level  # noqa: E402


def _ids(kind, B, vocab, rng):
    if kind == "uniform":
        return rng.integers(0, vocab, B)
    if kind == "hot":
        return np.where(rng.random(B) < 0.6, 7, rng.integers(0, vocab, B))
    if kind == "zipf":
        r = np.arange(1, vocab + 1, dtype=np.float64) ** -1.05
        return rng.choice(vocab, size=B, p=r / r.sum())
    if kind == "oob":
        x = rng.integers(-3, vocab + 3, B)
        return x
    return np.full(B, vocab - 1)


@pytest.mark.parametrize("kind", ["uniform", "hot", "zipf", "oob", "same"])
@pytest.mark.parametrize("B,P,chunk,fine", [(1000, 11, 256, 4), (4097, 43, 512, 4), (2048, 22, 2048, 12), (300, 5, 64, 1),
                                            (5000, 53, 4096, 16)])
def test_two_level_prepass_equals_the_direct_statement(kind, B, P, chunk, fine):
    rng = np.random.default_rng(B + P)
    vocab = 5000
    ids = _ids(kind, B, vocab, rng)
    c0, k0 = segments_direct(ids, vocab, P, bucket=64)
    c1, k1 = segments_two_level(ids, vocab, P, chunk, fine, bucket=64, rng=rng)
    assert np.array_equal(c0, c1)
    assert int(c0.sum()) == B
    for p in range(P):
        assert np.array_equal(k0[p], k1[p]), p
        if c0[p] <= 64:
            assert len(k0[p]) == c0[p] and np.all(np.diff(k0[p].astype(np.int64)) > 0)      # sorted, unique
    if kind in ("hot", "same"):
        assert (c0 > 64).any()            # the overflow rule is exercised
