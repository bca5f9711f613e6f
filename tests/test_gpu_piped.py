"""The software-pipelined screen (k_assign_screen_bf16_x32p: 8 tiles, sub_dim 8 / 16, row chunks of >= 8 steps) on the
cases the other suites only meet at 1M rows: row counts that are no multiple of 32 (a partial last step, clamped
loads), chunks of odd and even step counts (the dummy step), NaN / inf rows, rows next to ties, all three screened
metrics, codes through the transposing path (m >= 16) -- bit for bit against the oracle."""
import numpy as np
import pytest

import oracle as O
from vq_amd import _lib

pytestmark = pytest.mark.gpu
F = np.float32


def _codebook(rng, m, k, sd, ties=True):
    cb = rng.standard_normal((m, k, sd)).astype(F)
    if ties:
        for s in range(m):
            for j in range(0, k - 1, 8):  # every eighth centroid has a twin 1 ulp away in one coordinate
                cb[s, j + 1] = cb[s, j]
                t = (j // 8) % sd
                cb[s, j + 1, t] = np.nextafter(cb[s, j, t], F(9.0))
    return cb


def _rows(rng, n, cb):
    m, k, sd = cb.shape
    X = rng.standard_normal((n, m * sd)).astype(F)
    near = rng.integers(0, n, n // 50)  # rows on top of a centroid (the twins make them undecidable for the screen)
    for i in near:
        for s in range(m):
            X[i, s * sd:(s + 1) * sd] = cb[s, rng.integers(0, k)] + (1e-4 * rng.standard_normal(sd)).astype(F)
    X[5, 3] = np.nan
    X[n - 1, :] = np.inf          # the last row of a partial step
    X[n - 2, 7] = -np.inf
    X[n // 2, :] = 0.0            # |x| = 0: cosine's EPSILON rule
    X[n // 3, :] = np.nan
    return X


@pytest.mark.parametrize("shape", [(40_007, 8, 256, 16), (33_000, 8, 250, 16), (70_003, 16, 256, 8), (70_001, 4, 225, 16)])
@pytest.mark.parametrize("metric", [O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.COSINE])
def test_piped_encode_ragged_rows_bit_exact(oracle, shape, metric):
    n, m, k, sd = shape
    rng = np.random.default_rng(n + metric)
    cb = _codebook(rng, m, k, sd)
    X = _rows(rng, n, cb)
    enc = _lib.PQEncoder(cb, metric)
    codes, f16 = enc.encode(X)
    rechecked, engine = _lib.last_assign_stats()
    assert engine == _lib.ENGINE_MFMA_BF16 and rechecked > 0
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
    assert same.all()
    enc.close()


@pytest.mark.parametrize("shape", [(40_007, 8, 256, 16), (70_003, 16, 256, 8)])
def test_piped_lloyd_step_ragged_rows(oracle, shape):
    """The fused update of the pipelined kernel on a ragged row count with special rows: assignments and counts exact,
    centroids within the summation tolerance."""
    n, m, k, sd = shape
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, m * sd)).astype(F)
    X[n - 1, :] = 1e4   # a far row in the partial last step
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    km.init_from_rows(init)
    km.step()
    c_in = km.get_centroids()
    counts, changed = km.step()
    assert _lib.last_assign_stats()[1] == _lib.ENGINE_MFMA_BF16
    assign = km.get_assignments()
    c_out = km.get_centroids()
    for s in range(m):
        c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c_in[s], threads=0)
        assert int((assign[:, s].astype(np.uint32) != a_ref).sum()) == 0
        np.testing.assert_array_equal(counts[s], n_ref)
        assert bool(changed[s]) == ch_ref
        err = np.max(np.abs(c_out[s] - c1) / np.maximum(1.0, np.abs(c1)))
        assert err <= 1e-5, f"subspace {s}: centroid deviation {err:g}"
    km.close()
    ds.close()


@pytest.mark.parametrize("sd,m", [(16, 4), (8, 8)])
def test_piped_update_every_row_in_one_cluster(oracle, sd, m):
    """The fused update adds a proven row by LDS f64 atomics (round 6).  Worst case for them: EVERY row of a step goes to
    the same cluster -- 64 lanes of one atomic instruction on one address, serialised by the LDS.  Centroids far apart, all
    rows in a small ball around centroid 7 (every row proven by a wide margin): counts exact, the one populated cluster's
    mean within 2e-6 of the f64 mean of its rows (the reference's own sequential f32 sum of 120k terms is looser than that),
    twice the same bits."""
    n, k = 120_000, 256
    rng = np.random.default_rng(sd)
    cen = (rng.standard_normal((m, k, sd)) * 50.0).astype(F)
    X = np.concatenate([cen[s, 7][None, :] + 0.01 * rng.standard_normal((n, sd)).astype(F) for s in range(m)], axis=1).astype(F)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    outs = []
    for _ in range(2):
        km.set_centroids(cen)
        km.set_active(np.ones(m, np.uint8))
        counts, changed = km.step()
        assert _lib.last_assign_stats() == (0, _lib.ENGINE_MFMA_BF16)  # nothing re-checked: every row went through the atomics
        outs.append(km.get_centroids())
    np.testing.assert_array_equal(outs[0], outs[1])
    for s in range(m):
        assert counts[s, 7] == n and int(counts[s].sum()) == n
        want = X[:, s * sd:(s + 1) * sd].astype(np.float64).mean(axis=0)
        err = np.max(np.abs(outs[0][s, 7] - want) / np.maximum(1.0, np.abs(want)))
        assert err <= 2e-6, (s, err)
        others = np.delete(np.arange(k), 7)
        np.testing.assert_array_equal(outs[0][s, others], cen[s, others])  # empty clusters keep their centroids
    km.close()
    ds.close()


@pytest.mark.parametrize("tail_steps", [1, 2, 3, 4, 9])
@pytest.mark.parametrize("sd,m", [(16, 8), (8, 16)])
def test_piped_last_chunk_lengths(oracle, sd, m, tail_steps):
    """The last row chunk of a launch may be any number of steps long (the others are >= 8).  The two-step tail of the
    encode form pairs step S - 1 with step S: a chunk of one step has no partner for its only step (it is the dummy's
    partner after the loop), one of two or three steps ends on a real / a dummy second step -- codes bit for bit, and a
    partial last step on top."""
    waves = 1024 * (2 if sd == 8 else 1)
    chunks = waves // m
    per = 10                                            # steps per chunk (>= 8: the pipelined kernel)
    n_steps = (chunks - 1) * per + tail_steps           # the last chunk gets tail_steps steps
    n = n_steps * 32 - 13                               # ... the last one partial
    assert (n_steps + chunks - 1) // chunks == per
    rng = np.random.default_rng(100 * sd + tail_steps)
    cb = _codebook(rng, m, 256, sd)
    X = _rows(rng, n, cb)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    codes, _ = enc.encode(X, want_f16=False)
    assert _lib.last_assign_stats()[1] == _lib.ENGINE_MFMA_BF16
    want_c, _ = oracle.pq_encode(O.SQUARED_EUCLIDEAN, X, cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    enc.close()


def _fuzz_cases():
    import os

    scale = int(os.environ.get("VQ_FUZZ_SCALE", "1"))
    return list(range(12 * scale))


@pytest.mark.parametrize("seed", _fuzz_cases())
def test_piped_fuzz(oracle, seed):
    """Randomised shapes inside the pipelined kernel's range (k in 225..256, sub_dim 8 / 16, row chunks of >= 8 steps):
    row count, m, k, metric and data family drawn per seed; encode codes and f16 against the oracle."""
    rng = np.random.default_rng(9000 + seed)
    sd = int(rng.choice([8, 16]))
    m = int(rng.choice([4, 8, 16])) if sd == 16 else int(rng.choice([8, 16, 32]))
    k = int(rng.integers(225, 257))
    waves = 1024 * (2 if sd == 8 else 1)                 # one wave per SIMD, two at sub_dim 8
    n_min = 32 * 8 * max(1, waves // m) + 1              # >= 8 steps per row chunk
    n = int(rng.integers(n_min, n_min + 30_000))
    metric = int(rng.choice([O.SQUARED_EUCLIDEAN, O.EUCLIDEAN, O.COSINE]))
    kind = rng.choice(["uniform", "normal", "lattice", "clustered", "offset"])
    d = m * sd
    if kind == "uniform":
        X = rng.random((n, d), dtype=F)
    elif kind == "normal":
        X = rng.standard_normal((n, d)).astype(F)
    elif kind == "lattice":
        X = rng.integers(0, 3, (n, d)).astype(F)
    elif kind == "clustered":
        centers = rng.standard_normal((40, d)).astype(F) * 3
        X = (centers[rng.integers(0, 40, n)] + 0.05 * rng.standard_normal((n, d))).astype(F)
    else:
        X = (rng.random((n, d), dtype=F) + F(1000.0)).astype(F)
    cb = X[rng.choice(n, k, replace=False)].reshape(k, m, sd).transpose(1, 0, 2).copy()  # centroids = rows: exact hits and ties
    cb += (1e-3 * rng.standard_normal(cb.shape)).astype(F) * (rng.random(cb.shape) < 0.5)
    enc = _lib.PQEncoder(cb, metric)
    codes, f16 = enc.encode(X)
    assert _lib.last_assign_stats()[1] == _lib.ENGINE_MFMA_BF16
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)
    same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
    assert same.all()
    enc.close()
