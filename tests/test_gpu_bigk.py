"""GPU parity for codebooks of more than 256 centroids (two-byte codes, include/vqhip.h "code width").

The reference's `k` is a plain usize (src/pq.rs:83-96, src/core/vector.rs:390-406) and its best_idx a
usize; above 256 the library switches every codes buffer to u16; the bf16 X32 screen serves it with up to 16
centroid groups per subspace (merged per row), the exact engine everything else.  Same bar as
test_gpu_parity.py: codes, counts and f16 outputs bit-exact against the oracle; centroids bit-exact with
the reference-order update and within CENTROID_RTOL with the blocked one.
"""
import numpy as np
import pytest

import oracle as O
from vq_amd import _lib
from vq_amd.pq import ProductQuantizer, fit_codebooks
from vq_amd.distance import Distance

pytestmark = pytest.mark.gpu

F = np.float32
CENTROID_RTOL = 1e-5


def _data(seed, n, d, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d), dtype=F)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(F)
    if kind == "lattice":
        return rng.integers(0, 4, (n, d)).astype(F)
    raise ValueError(kind)


SHAPES = [  # (n, d, m, k)
    (1500, 64, 4, 257),     # first two-byte k
    (1200, 128, 8, 300),
    (2000, 32, 2, 512),
    (1100, 48, 4, 1000),    # sub_dim 12
    (900, 10, 2, 333),      # sub_dim 5: generic exact kernel
    (1300, 64, 1, 1024),    # m = 1, sub_dim 64
    (700, 16, 4, 4096),     # sub_dim 4: no X32 form -> exact engine
    (1000, 64, 2, 600),     # sub_dim 32: 5 groups of 128
    (800, 96, 2, 500),      # sub_dim 48: 8 groups of 64
    (900, 48, 2, 2000),     # sub_dim 24: 8 groups of 256 (last one ragged)
    (600, 16, 2, 4096),     # sub_dim 8: 16 groups, the most the screen takes
    (600, 16, 2, 4100),     # one more group than that -> exact engine
]


def _x32_groups(sd, k):
    cap = {8: 8, 12: 8, 16: 8, 24: 8, 32: 4, 48: 2, 64: 2}.get(sd)
    if cap is None:
        return 0
    nt = (k + 31) // 32
    g = (nt + min(nt, cap) - 1) // min(nt, cap)
    return g if g <= 16 else 0


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN, O.COSINE])
@pytest.mark.parametrize("engine", [_lib.ENGINE_AUTO, _lib.ENGINE_EXACT])
def test_encode_bit_exact(oracle, shape, kind, metric, engine):
    n, d, m, k = shape
    X = _data(1, n, d, kind)
    cb = _data(2, m * k, d // m, kind).reshape(m, k, d // m)
    if kind == "lattice":
        cb[:, k // 2] = cb[:, 0]  # duplicate centroid: the lower index must win
    enc = _lib.PQEncoder(cb, metric)
    enc.set_engine(engine)
    codes, f16 = enc.encode(X)
    assert codes.dtype == np.uint16 and codes.shape == (n, m)
    rechecked, used = _lib.last_assign_stats()
    # AUTO: the screen where the shape has one -- unless the whole pass is so little work that the one-launch exact scan wins
    screened = (engine == _lib.ENGINE_AUTO and metric != O.MANHATTAN and _x32_groups(d // m, k) > 0 and n * m * k * (d // m) > 32e6)
    if _lib.selftest()[2]:
        assert used == (_lib.ENGINE_MFMA_BF16 if screened else _lib.ENGINE_EXACT)
    if used == _lib.ENGINE_MFMA_BF16 and kind == "uniform" and metric != O.COSINE:
        assert rechecked < 0.2 * n * m  # the screen decides most rows by itself
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
    assert same.all()
    # codes that need the high byte must occur, or the test proves nothing
    assert int(codes.max()) > 255
    # decode reads the same two-byte codes back
    rec = enc.decode(codes)
    want_rec = np.concatenate([cb[s][want_c[:, s]] for s in range(m)], axis=1)
    assert rec.tobytes() == want_rec.tobytes()
    # the per-vector latency path (n <= 8) writes the same width
    few, few16 = enc.encode(X[:5])
    assert few.dtype == np.uint16
    np.testing.assert_array_equal(few, codes[:5])
    assert few16.view(np.uint16).tobytes() == f16[:5].view(np.uint16).tobytes()
    enc.close()


def test_screen_engines_without_a_wide_form_refuse():
    # fp32 MFMA screen: k <= 256 only; bf16 X32: sub_dim 20 has no instantiation, k = 4100 at sub_dim 8 too many groups
    for sd, k, eng in ((16, 300, _lib.ENGINE_MFMA), (20, 300, _lib.ENGINE_MFMA_BF16), (8, 4100, _lib.ENGINE_MFMA_BF16)):
        cb = _data(3, 2 * k, sd, "uniform").reshape(2, k, sd)
        enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
        enc.set_engine(eng)
        with pytest.raises(_lib.FfiError):
            enc.encode(_data(4, 100, 2 * sd, "uniform"))
        enc.close()


@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN])
@pytest.mark.parametrize("shape", [(5000, 16, 4, 300), (3000, 32, 2, 1024), (2500, 8, 8, 600)])
def test_adc_search_over_two_byte_codes(oracle, shape, metric):
    """the code-based search reads u16 codes above 256 centroids; fewer query tables share the LDS per pass"""
    n, d, m, k = shape
    cb = _data(5, m * k, d // m, "normal").reshape(m, k, d // m)
    X = _data(6, n, d, "normal")
    enc = _lib.PQEncoder(cb, metric)
    codes, _ = enc.encode(X, want_f16=False)
    assert codes.dtype == np.uint16
    Q = _data(7, 19, d, "normal")
    idx, dist = enc.adc_search(codes, Q, 10)
    want_i, want_d = oracle.adc_search(metric, cb, codes, Q, 10)
    np.testing.assert_array_equal(idx, want_i)
    assert dist.tobytes() == want_d.tobytes()
    enc.close()


def test_decode_rejects_code_beyond_k():
    cb = _data(6, 300 * 8, 1, "uniform").reshape(1, 300, 8)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    with pytest.raises(_lib.FfiError):
        enc.decode(np.array([[300]], np.uint16))
    enc.close()


STEP_SHAPES = [
    (6000, 64, 4, 300),
    (5000, 32, 2, 1024),
    (9000, 48, 4, 512),    # sub_dim 12
    (4000, 10, 2, 333),    # generic accumulate kernel
    (12000, 32, 2, 4096),  # k*(sub_dim+2) = 73728 words > one CU's LDS: two cluster ranges per subspace
    (8000, 128, 1, 700),   # sub_dim 128: 91000 words, three ranges
]


@pytest.mark.parametrize("shape", STEP_SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "normal"])
def test_lloyd_step_parity(oracle, shape, kind):
    n, d, m, k = shape
    sd = d // m
    X = _data(11, n, d, kind)
    init = np.array([[(j * (n // k) + 7 * s) % n for j in range(k)] for s in range(m)], np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(init)
    counts, changed = km.step()
    assign = km.get_assignments()
    assert assign.dtype == np.uint16
    cent = km.get_centroids()
    for s in range(m):
        c0 = X[init[s].astype(np.int64), s * sd:(s + 1) * sd]
        c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c0, threads=0)
        np.testing.assert_array_equal(assign[:, s].astype(np.uint32), a_ref)
        np.testing.assert_array_equal(counts[s], n_ref)
        nonempty = n_ref > 0
        err = np.abs(cent[s][nonempty] - c1[nonempty]) / np.maximum(1.0, np.abs(c1[nonempty]))
        assert err.max() <= CENTROID_RTOL
        np.testing.assert_array_equal(cent[s][~nonempty], c0[~nonempty])
        assert bool(changed[s]) == ch_ref
    km.close()
    ds.close()


@pytest.mark.parametrize("shape", STEP_SHAPES)
def test_exact_update_step_is_bit_identical(oracle, shape):
    n, d, m, k = shape
    sd = d // m
    X = _data(31, n, d, "uniform")
    init = np.array([[(j * (n // k) + 3 * s) % n for j in range(k)] for s in range(m)], np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.set_exact_update(True)
    km.init_from_rows(init)
    counts, changed = km.step()
    cent = km.get_centroids()
    for s in range(m):
        c0 = X[init[s].astype(np.int64), s * sd:(s + 1) * sd]
        c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c0, threads=0)
        np.testing.assert_array_equal(counts[s], n_ref)
        nonempty = n_ref > 0
        assert cent[s][nonempty].tobytes() == c1[nonempty].tobytes()
        assert bool(changed[s]) == ch_ref
    km.close()
    ds.close()


def test_exact_update_full_fit_is_bit_identical(oracle):
    n, d, m, k = 12000, 32, 2, 400
    X = _data(32, n, d, "uniform")
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    init[:, 1] = init[:, 0]  # duplicate initial centroid -> empty cluster -> reseed path
    reseed = np.array([[11, 222, 3333, 4444, 5555, 6666, 7777, 8888]] * m, np.uint64)
    ds = _lib.Dataset.from_host(X)
    stats = {}
    cb = fit_codebooks(ds, m, k, 6, init_rows=init, reseed_rows=reseed, exact_update=True, stats=stats)
    ds.close()
    cb_ref, it_ref = oracle.pq_fit(X, m, k, 6, init, reseed_rows=reseed, threads=0)
    assert stats["iters"].tolist() == it_ref.tolist()
    assert cb.tobytes() == cb_ref.tobytes()


def test_product_quantizer_front_end_wide_k(oracle):
    """the reference-shaped class end to end: fit (host RNG init), quantize, encode, decode"""
    n, d, m, k = 5000, 32, 4, 320
    X = _data(41, n, d, "uniform")
    pq = ProductQuantizer(X, m, k, 3, Distance.squared_euclidean(), 7)
    cb = pq.codebooks
    assert cb.shape == (m, k, d // m)
    codes = pq.encode(X[:1000])
    assert codes.dtype == np.uint16
    want_c, want_f = oracle.pq_encode(O.SQUARED_EUCLIDEAN, X[:1000], cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    q = pq.quantize(X[0])
    assert q.view(np.uint16).tobytes() == want_f[0].tobytes()
    assert pq.decode(codes).tobytes() == np.concatenate([cb[s][want_c[:, s]] for s in range(m)], axis=1).tobytes()
    idx, dist = pq.search(codes, X[:2], 5)
    want_i, want_d = oracle.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, X[:2], 5)
    np.testing.assert_array_equal(idx, want_i)


def test_code_bytes_entry_point():
    lib = _lib.load()
    assert [lib.vqhip_code_bytes(k) for k in (1, 256, 257, 65536)] == [1, 1, 2, 2]
