"""Host-side pieces next to the hot path: the code-based index file (vq_amd/store.py) and the
evaluation report helpers restating src/bin/common.rs (vq_amd/evalcli.py).  CPU only."""
import numpy as np
import pytest

from vq_amd.distance import Distance
from vq_amd.evalcli import recall_at_k, reconstruction_error
from vq_amd.store import PQIndex

F = np.float32


def _index(n=500, m=4, k=16, sd=3, seed=0):
    rng = np.random.default_rng(seed)
    return PQIndex(rng.standard_normal((m, k, sd)).astype(F), rng.integers(0, k, (n, m)).astype(np.uint8),
                   Distance.cosine())


def test_index_roundtrip_and_mmap(tmp_path):
    idx = _index()
    p = tmp_path / "a.vqpq"
    idx.save(p)
    assert p.stat().st_size == idx.nbytes == 32 + 4 * 16 * 3 * 4 + 500 * 4
    for mm in (False, True):
        back = PQIndex.load(p, mmap_codes=mm)
        np.testing.assert_array_equal(back.codebooks, idx.codebooks)
        np.testing.assert_array_equal(np.asarray(back.codes), idx.codes)
        assert back.distance == Distance.cosine() and (back.m, back.k, back.dim, len(back)) == (4, 16, 12, 500)


def test_index_with_more_than_256_centroids_keeps_two_byte_codes(tmp_path):
    rng = np.random.default_rng(5)
    m, k, sd, n = 3, 700, 2, 400
    codes = rng.integers(0, k, (n, m))
    codes[0] = k - 1
    idx = PQIndex(rng.standard_normal((m, k, sd)).astype(F), codes)
    assert idx.codes.dtype == np.dtype("<u2")
    p = tmp_path / "wide.vqpq"
    idx.save(p)
    assert p.stat().st_size == idx.nbytes == 32 + m * k * sd * 4 + n * m * 2
    for mm in (False, True):
        back = PQIndex.load(p, mmap_codes=mm)
        np.testing.assert_array_equal(np.asarray(back.codes), codes)
        np.testing.assert_array_equal(back.reconstruct([0])[0], np.concatenate([idx.codebooks[s][k - 1] for s in range(m)]))
    with pytest.raises(ValueError, match="out of range"):
        PQIndex(idx.codebooks, np.full((2, m), k))


def test_index_reconstruct_is_the_reference_quantize_output():
    idx = _index()
    rec = idx.reconstruct()
    for i in (0, 17, 499):
        want = np.concatenate([idx.codebooks[s][idx.codes[i, s]] for s in range(idx.m)])
        np.testing.assert_array_equal(rec[i], want)
    np.testing.assert_array_equal(idx.reconstruct_f16([3, 5]), rec[[3, 5]].astype(np.float16))


def test_index_rejects_bad_input(tmp_path):
    rng = np.random.default_rng(1)
    cb = rng.random((2, 4, 3), dtype=F)
    with pytest.raises(ValueError, match="shape"):
        PQIndex(cb, np.zeros((5, 3), np.uint8))
    with pytest.raises(ValueError, match="out of range"):
        PQIndex(cb, np.full((5, 2), 4, np.uint8))
    p = tmp_path / "bad"
    p.write_bytes(b"nope" * 20)
    with pytest.raises(ValueError, match="VQPQIDX1"):
        PQIndex.load(p)
    idx = PQIndex(cb, np.zeros((5, 2), np.uint8))
    idx.save(p)
    p.write_bytes(p.read_bytes()[:-3])
    with pytest.raises(ValueError, match="truncated"):
        PQIndex.load(p)


def test_reconstruction_error_matches_common_rs_formula():
    # src/bin/common.rs:61-78
    a = np.array([[1, 2], [3, 4]], F)
    b = np.array([[1, 1], [5, 4]], F)
    assert reconstruction_error(a, b) == pytest.approx((0 + 1 + 4 + 0) / 4)


def _recall_loops(original, approx, k):
    """literal restatement of calculate_recall, src/bin/common.rs:91-130"""
    n = len(original)
    step = max(n // min(n, 1000), 1)
    total = 0.0
    for i in range(0, n, step):
        window = 5000 if n > 10_000 else n
        lo, hi = max(i - window // 2, 0), min(i + window // 2, n)
        js = [j for j in range(lo, hi) if j != i]
        t = sorted(js, key=lambda j: float(((original[i] - original[j]) ** 2).sum()))[:k]
        a = sorted(js, key=lambda j: float(((approx[i] - approx[j]) ** 2).sum()))[:k]
        total += len(set(t) & set(a)) / k
    return total / (n // step)


def test_recall_matches_common_rs_protocol():
    rng = np.random.default_rng(2)
    X = rng.random((240, 6), dtype=F)
    A = (X * 8).round() / 8
    assert recall_at_k(X, X, 10) == 1.0
    assert recall_at_k(X, A, 10) == pytest.approx(_recall_loops(X, A, 10), abs=1e-12)
    assert 0.0 < recall_at_k(X, A, 5) < 1.0
