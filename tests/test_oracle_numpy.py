"""Cross-checks the C oracle against the independent numpy-scalar restatement
(tests/ref_numpy.py) on small seeded inputs: bit-exact in every case."""
import numpy as np
import pytest

import oracle as O
import ref_numpy as R

F = np.float32


def _data(seed, n, d, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d), dtype=F)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(F)
    if kind == "lattice":  # many exact ties
        return rng.integers(0, 3, (n, d)).astype(F)
    if kind == "wide":  # wide dynamic range incl. denormals
        e = rng.integers(-45, 20, (n, d))
        return (rng.standard_normal((n, d)) * (10.0 ** e)).astype(F)
    raise ValueError(kind)


@pytest.mark.parametrize("metric", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice", "wide"])
def test_distance_bit_exact(oracle, metric, kind):
    x = _data(1, 60, 16, kind)
    for i in range(0, 60, 2):
        a, b = x[i], x[i + 1]
        got, want = oracle.distance(metric, a, b), R.distance(metric, a, b)
        assert got.tobytes() == F(want).tobytes() or (np.isnan(got) and np.isnan(want))
    for i in range(0, 60, 2):
        assert oracle.distance2(x[i], x[i + 1]).tobytes() == R.distance2(x[i], x[i + 1]).tobytes()


@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
def test_lloyd_bit_exact(oracle, kind):
    data = _data(2, 120, 4, kind)
    k = 6
    init = [3, 17, 40, 41, 99, 100]
    reseed = [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    want, it_w = R.lloyd(data, k, 8, init, reseed)
    got, it_g, _ = oracle.lloyd(data, k, 8, init, reseed)
    assert it_w == it_g
    assert got.tobytes() == want.tobytes()
    # strided view of a wider matrix == the contiguous copy (pq.rs:122-129 slicing)
    wide = np.concatenate([_data(3, 120, 4, kind), data, _data(4, 120, 4, kind)], axis=1)
    got2, it2, _ = oracle.lloyd(wide[:, 4:8], k, 8, init, reseed)
    assert it2 == it_g and got2.tobytes() == got.tobytes()
    # threads change nothing
    got3, it3, _ = oracle.lloyd(data, k, 8, init, reseed, threads=4)
    assert it3 == it_g and got3.tobytes() == got.tobytes()


def test_lloyd_empty_cluster_reseed_order(oracle):
    # duplicates force empty clusters: init rows 0 and 1 are identical points, so cluster 1
    # never wins a strict '<' and is reseeded every iteration (vector.rs:448-452)
    data = np.array([[0, 0], [0, 0], [1, 1], [5, 5], [5, 6], [9, 9]], F)
    reseed = [5, 4, 3, 2]
    want, it_w = R.lloyd(data, 3, 4, [0, 1, 3], reseed)
    got, it_g, used = oracle.lloyd(data, 3, 4, [0, 1, 3], reseed)
    assert it_w == it_g and got.tobytes() == want.tobytes()
    assert used >= 1
    with pytest.raises(O.OracleError) as e:
        oracle.lloyd(data, 3, 4, [0, 1, 3], [])
    assert e.value.code == O.ERR_RESEED_EXHAUSTED


@pytest.mark.parametrize("metric", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
def test_pq_encode_bit_exact(oracle, metric, kind):
    rows = _data(5, 40, 8, kind)
    cb = _data(6, 2 * 5, 4, kind).reshape(2, 5, 4)
    if kind == "lattice":
        cb[0, 3] = cb[0, 1]  # duplicate centroid: the lower index must win
        cb[1, 0] = 0  # zero centroid: cosine -> 1.0
    wc, wf = R.pq_encode(metric, rows, cb)
    gc, gf = oracle.pq_encode(metric, rows, cb)
    np.testing.assert_array_equal(gc, wc)
    np.testing.assert_array_equal(gf, wf.view(np.uint16))


def test_pq_encode_nan_centroid_zero_blocks_all(oracle):
    # if centroid 0's distance is NaN no later `dist < best` is ever true (pq.rs:187)
    rows = np.array([[1, 2, 3, 4]], F)
    cb = np.array([[[np.nan, 0], [1, 2]], [[3, 4], [np.nan, 1]]], F)
    wc, _ = R.pq_encode(0, rows, cb)
    gc, _ = oracle.pq_encode(0, rows, cb)
    np.testing.assert_array_equal(gc, wc)
    assert gc.tolist() == [[0, 0]]


@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
@pytest.mark.parametrize("depth", [0, 1, 3, 6])
def test_tsvq_bit_exact(oracle, kind, depth):
    rows = _data(7, 70, 5, kind)
    if kind == "normal":
        rows[11, 2] = np.nan  # partial NaN (regression_tests.rs:282-297)
    want = R.tsvq_build(rows, depth)
    wc, wl, wr = R.tsvq_flatten(want)
    got = oracle.tsvq_build(rows, depth)
    np.testing.assert_array_equal(got["left"], wl)
    np.testing.assert_array_equal(got["right"], wr)
    assert got["centroids"].tobytes() == wc.tobytes() or (
        np.array_equal(np.isnan(got["centroids"]), np.isnan(wc))
        and np.array_equal(np.nan_to_num(got["centroids"]), np.nan_to_num(wc)))
    q = _data(8, 25, 5, kind)
    for metric in (0, 1, 2, 3):
        leaf, f16 = oracle.tsvq_encode(metric, q, got)
        for i in range(q.shape[0]):
            nd = R.tsvq_find_leaf(metric, want, q[i])
            exp = nd["centroid"].astype(np.float16).view(np.uint16)
            act = f16[i]
            assert np.array_equal(act, exp) or np.isnan(nd["centroid"]).any()
