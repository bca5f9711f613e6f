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
