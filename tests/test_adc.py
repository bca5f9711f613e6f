"""Asymmetric-distance search over stored codes.  The reference has no such
function, so the oracle's vqo_adc_search DEFINES the semantics (built from the restated distance
kernels); here: the oracle against a plain numpy statement of the same definition (CPU), and the
GPU path against the oracle, bit for bit (indices and distance bits, ties by row)."""
import numpy as np
import pytest

import oracle as O

F = np.float32


def _numpy_adc(metric, cb, codes, queries, topk):
    m, k, sd = cb.shape
    out_i, out_d = [], []
    for q in queries:
        acc = None
        for s in range(m):
            diff = (q[s * sd:(s + 1) * sd][None, :] - cb[s]).astype(F)
            if metric == O.MANHATTAN:
                t = np.zeros(k, F)
                for c in range(sd):
                    t = (t + np.abs(diff[:, c])).astype(F)
            else:
                t = np.zeros(k, F)
                for c in range(sd):
                    t = (t + (diff[:, c] * diff[:, c]).astype(F)).astype(F)
            term = t[codes[:, s]]
            acc = term if acc is None else (acc + term).astype(F)
        order = np.lexsort((np.arange(len(acc)), np.where(np.isnan(acc), np.inf, acc)))[:topk]
        out_i.append(order)
        out_d.append(np.sqrt(acc[order]) if metric == O.EUCLIDEAN else acc[order])
    return np.array(out_i, np.uint32), np.array(out_d, F)


@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN])
def test_oracle_adc_is_the_stated_definition(oracle, metric):
    rng = np.random.default_rng(5)
    cb = rng.standard_normal((4, 16, 3)).astype(F)
    codes = rng.integers(0, 16, (300, 4)).astype(np.uint8)
    codes[10] = codes[20] = codes[5]  # exact ties: lower row first
    Q = rng.standard_normal((6, 12)).astype(F)
    idx, dist = oracle.adc_search(metric, cb, codes, Q, 25)
    want_i, want_d = _numpy_adc(metric, cb, codes, Q, 25)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    # and it is the distance to the decoded row up to the different summation grouping
    dec = np.concatenate([cb[s][codes[:, s]] for s in range(4)], axis=1)
    full = ((Q[0][None, :] - dec) ** 2).sum(axis=1) if metric != O.MANHATTAN else np.abs(Q[0][None, :] - dec).sum(axis=1)
    ref = np.sqrt(full[idx[0]]) if metric == O.EUCLIDEAN else full[idx[0]]
    np.testing.assert_allclose(dist[0], ref, rtol=1e-5)


def test_oracle_adc_rejects_cosine(oracle):
    with pytest.raises(O.OracleError):
        oracle.adc_search(O.COSINE, np.zeros((2, 4, 2), F), np.zeros((5, 2), np.uint8), np.zeros((1, 4), F), 1)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN])
@pytest.mark.parametrize("shape", [(5000, 8, 16, 256, 10, 9), (70_001, 16, 8, 200, 100, 3), (300, 4, 4, 7, 300, 1),
                                   (20_000, 2, 24, 256, 1024, 17), (1, 3, 5, 2, 1, 2),
                                   (30_000, 8, 8, 64, 20, 70)])  # 70 queries: a full group of 64 (eight scan batches in one set of launches) + 6
def test_gpu_adc_search_bit_exact(oracle, metric, shape):
    from vq_amd import _lib

    n, m, sd, k, topk, nq = shape
    rng = np.random.default_rng(n + m)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8)
    if n > 50:
        codes[n // 2: n // 2 + 20] = codes[3]  # blocks of exact ties across the k-th place
    Q = rng.standard_normal((nq, m * sd)).astype(F)
    enc = _lib.PQEncoder(cb, metric)
    idx, dist = enc.adc_search(codes, Q, topk)
    want_i, want_d = oracle.adc_search(metric, cb, codes, Q, topk)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    enc.close()


@pytest.mark.gpu
def test_gpu_adc_lattice_ties_nan_and_host_mirror(oracle, tmp_path):
    import vq_amd as pyvq
    from vq_amd.store import PQIndex

    rng = np.random.default_rng(9)
    X = rng.integers(0, 3, (4000, 16)).astype(F)  # lattice: massive ties
    pq = pyvq.ProductQuantizer(X, 4, 8, max_iters=4, distance=pyvq.Distance.squared_euclidean())
    codes = pq.encode(X)
    idx, dist = pq.search(codes, X[:5], topk=50)
    want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, pq.codebooks, codes, X[:5], 50)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist, want_d)
    assert (np.diff(dist, axis=1) >= 0).all()
    index = PQIndex.from_quantizer(pq, X)
    i2, d2 = index.search(X[:5], 50)
    np.testing.assert_array_equal(i2, idx)
    # NaN query component: every distance NaN -> rows in index order
    q = X[:1].copy()
    q[0, 3] = np.nan
    i3, d3 = pq.search(codes, q, topk=7)
    w3, _ = oracle.adc_search(O.SQUARED_EUCLIDEAN, pq.codebooks, codes, q, 7)
    np.testing.assert_array_equal(i3, w3)
    assert np.isnan(d3).all()
    with pytest.raises(pyvq.InvalidParameter, match="cosine"):
        pyvq.ProductQuantizer(X, 4, 8, max_iters=1, distance=pyvq.Distance.cosine()).search(codes, X[:1], 3)
    with pytest.raises(pyvq.DimensionMismatch):
        pq.search(codes, np.zeros((1, 5), F), 3)


@pytest.mark.gpu
def test_gpu_adc_dense_ties_take_the_radix_select_path(oracle):
    """More than 8192 rows tie below the histogram cut (identical codes): the candidate filter hands
    the query to the exact radix select; ties still resolve by row index."""
    from vq_amd import _lib

    rng = np.random.default_rng(21)
    m, k, sd, n = 4, 32, 4, 30_000
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = np.tile(rng.integers(0, k, (1, m)).astype(np.uint8), (n, 1))
    far = rng.choice(n, 500, replace=False)
    codes[far] = rng.integers(0, k, (500, m)).astype(np.uint8)
    Q = np.concatenate([cb[s][codes[0, s]] for s in range(m)])[None, :] + F(0.01)  # nearest = the repeated code
    Q = np.concatenate([Q, rng.standard_normal((2, m * sd)).astype(F)])
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    idx, dist = enc.adc_search(codes, Q, 300)
    want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, 300)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    enc.close()


# ---- the one-scan schedule (n >= 32768, topk <= 256): a sampled threshold, candidates only ----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN])
@pytest.mark.parametrize("shape", [(200_000, 8, 16, 256, 10, 64),      # the bench's shape at a fifth of its rows: 8 batches of 8 queries
                                   (1_000_000, 8, 16, 256, 10, 9),     # full size; the last batch holds one query
                                   (50_000, 16, 8, 200, 256, 5),       # the largest topk the schedule takes
                                   (40_000, 96, 8, 256, 20, 3),        # C3's tables: 98 KB each, one query per batch
                                   (33_000, 3, 5, 7, 40, 11),          # m not a multiple of 8: the byte-load path; heavy ties (7^3 distinct rows)
                                   (100_000, 4, 4, 300, 7, 6)])        # two-byte codes
def test_gpu_adc_one_scan_bit_exact(oracle, metric, shape):
    from vq_amd import _lib

    n, m, sd, k, topk, nq = shape
    rng = np.random.default_rng(n + m + 1)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8 if k <= 256 else np.uint16)
    codes[n // 2: n // 2 + 20] = codes[3]
    Q = rng.standard_normal((nq, m * sd)).astype(F)
    Q[0] = cb[np.arange(m), codes[12345].astype(np.int64)].reshape(-1)  # a query that IS a stored row: D = 0 at the top
    enc = _lib.PQEncoder(cb, metric)
    idx, dist = enc.adc_search(codes, Q, topk)
    redone = enc.adc_last_redone()
    want_i, want_d = oracle.adc_search(metric, cb, codes, Q, topk)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    # the threshold is placed for ~1024-4096 candidates: on random codes no query should need the full pass, except where
    # ties pile more than 8192 rows onto one value (the 7^3-row shape)
    if k ** m > 10 * n:
        assert redone == 0, redone
    enc.close()


@pytest.mark.gpu
def test_gpu_adc_flagged_queries_take_the_full_pass(oracle):
    """every query flagged by the test hook: the caller's repeat path must give the same bits"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path[:0] = [".", "oracle"]
        import numpy as np, oracle as O
        from vq_amd import _lib
        F = np.float32
        rng = np.random.default_rng(3)
        n, m, sd, k, topk, nq = 60_000, 8, 4, 64, 15, 10
        cb = rng.standard_normal((m, k, sd)).astype(F)
        codes = rng.integers(0, k, (n, m)).astype(np.uint8)
        Q = rng.standard_normal((nq, m * sd)).astype(F)
        enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
        idx, dist = enc.adc_search(codes, Q, topk)
        assert enc.adc_last_redone() == nq, enc.adc_last_redone()
        orc = O.get()
        wi, wd = orc.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, topk)
        assert (idx == wi).all() and (dist.view(np.uint32) == wd.view(np.uint32)).all()
        print("ok")
    """)
    import os
    env = dict(os.environ, VQHIP_TEST_ADC_REDO="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_gpu_adc_constant_distances_fall_back(oracle):
    """all rows equal: every D(q, .) ties, the threshold passes n > 8192 rows, the full pass answers (lowest rows first)"""
    from vq_amd import _lib

    n, m, sd, k, topk, nq = 40_000, 8, 2, 16, 12, 3
    rng = np.random.default_rng(9)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = np.tile(rng.integers(0, k, (1, m)).astype(np.uint8), (n, 1))
    Q = rng.standard_normal((nq, m * sd)).astype(F)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    idx, dist = enc.adc_search(codes, Q, topk)
    assert enc.adc_last_redone() == nq
    np.testing.assert_array_equal(idx, np.tile(np.arange(topk, dtype=np.uint32), (nq, 1)))
    want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, topk)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40_000, 4, 4, 16, 5, 5000),     # more queries than one set of launches takes (4096)
                                   (40_000, 8, 64, 32, 9, 600)])    # 1.2 MB of queries: the copies instead of the pinned stage
def test_gpu_adc_one_scan_many_queries(oracle, shape):
    from vq_amd import _lib

    n, m, sd, k, topk, nq = shape
    rng = np.random.default_rng(nq)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8)
    Q = rng.standard_normal((nq, m * sd)).astype(F)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    idx, dist = enc.adc_search(codes, Q, topk)
    want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, topk)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16 * int(__import__("os").environ.get("VQ_FUZZ_SCALE", "1"))))
def test_gpu_adc_one_scan_random_shapes(oracle, seed):
    """random shapes above the one-scan schedule's row floor: table widths that are not powers of two, topk on both sides of
    the wave-level top-k's limit (64), few distinct rows (long tie runs across the k-th place), duplicated blocks"""
    from vq_amd import _lib

    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(32_768, 120_000))
    m = int(rng.integers(1, 13))
    k = int(rng.choice([2, 3, 16, 37, 255, 256, 257, 300]))
    sd = int(rng.integers(1, 9))
    topk = int(rng.choice([1, 2, 10, 63, 64, 65, 200, 256]))
    nq = int(rng.integers(1, 21))
    metric = [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN][seed % 3]
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8 if k <= 256 else np.uint16)
    if seed % 4 == 0:
        codes[n // 3: n // 3 + 300] = codes[7]          # 300 exact ties
    if seed % 5 == 0:
        codes[:, : max(1, m // 2)] = codes[0, : max(1, m // 2)]  # half the subspaces constant: fewer distinct distances
    Q = rng.standard_normal((nq, m * sd)).astype(F)
    enc = _lib.PQEncoder(cb, metric)
    idx, dist = enc.adc_search(codes, Q, topk)
    want_i, want_d = oracle.adc_search(metric, cb, codes, Q, topk)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    enc.close()


@pytest.mark.gpu
def test_gpu_adc_one_scan_nan_and_inf_queries(oracle):
    """a NaN query component makes every distance NaN (no row is <= any threshold: the full pass answers, rows in index
    order); a huge component overflows every distance to +inf (all rows tie at +inf)"""
    from vq_amd import _lib

    n, m, sd, k, topk = 50_000, 8, 4, 64, 6
    rng = np.random.default_rng(77)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8)
    Q = rng.standard_normal((4, m * sd)).astype(F)
    Q[1, 5] = np.nan
    Q[2, 9] = F(3e38)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    idx, dist = enc.adc_search(codes, Q, topk)
    want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, topk)
    np.testing.assert_array_equal(idx, want_i)
    assert np.isnan(dist[1]).all() and np.isnan(want_d[1]).all()
    keep = [0, 2, 3]
    np.testing.assert_array_equal(dist[keep].view(np.uint32), want_d[keep].view(np.uint32))
    assert enc.adc_last_redone() == 2
    enc.close()


@pytest.mark.gpu
def test_gpu_adc_resident_code_store(oracle, tmp_path):
    """PQIndex keeps its codes on the device after the first search; an out-of-range code is refused at upload"""
    from vq_amd import _lib
    from vq_amd.store import PQIndex
    import vq_amd as pyvq

    rng = np.random.default_rng(31)
    n, m, sd, k = 45_000, 8, 4, 64
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8)
    index = PQIndex(cb, codes, pyvq.Distance.squared_euclidean())
    for rep in range(3):
        Q = rng.standard_normal((5 + rep, m * sd)).astype(F)
        idx, dist = index.search(Q, 12)
        want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, 12)
        np.testing.assert_array_equal(idx, want_i)
        np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
    with pytest.raises(pyvq.DimensionMismatch):
        index.search(np.zeros((1, 3), F), 3)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    with pytest.raises(_lib.FfiError, match="no codes loaded"):
        enc.adc_search(None, Q, 3)
    bad = codes.copy()
    bad[17, 3] = k
    with pytest.raises(_lib.FfiError, match="outside"):
        enc.adc_set_codes(bad)
    enc.adc_set_codes(codes[:40_000])
    i2, d2 = enc.adc_search(None, Q, 4)
    w2, wd2 = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes[:40_000], Q, 4)
    np.testing.assert_array_equal(i2, w2)
    np.testing.assert_array_equal(d2.view(np.uint32), wd2.view(np.uint32))
    enc.close()
