"""Pins the CPU oracle against every RNG-independent known-answer case the reference's own
tests hold for the k-means / nearest-centroid path.  The reference
(Rust) cannot be built here and ships no golden vectors, so these are the strongest pins
available; each case cites the reference test it restates.
"""
import math

import numpy as np
import pytest

import oracle as O

F = np.float32


def test_dot_norm_distance2_mean(oracle):
    # src/core/vector.rs:505-538 and tests/regression_tests.rs:157-163 (dot == 32.0 exactly)
    assert oracle.dot([1, 2, 3], [4, 5, 6]) == F(32.0)
    assert oracle.norm([3, 4]) == F(5.0)
    assert oracle.distance2([1, 2, 3], [4, 5, 6]) == F(27.0)
    np.testing.assert_array_equal(
        oracle.mean_vector([[1, 2, 3], [4, 5, 6], [7, 8, 9]]), np.array([4, 5, 6], F)
    )


def test_mean_vector_empty_is_error(oracle):
    # src/core/vector.rs:540-545
    with pytest.raises(O.OracleError) as e:
        oracle.mean_vector(np.zeros((0, 3), F))
    assert e.value.code == O.ERR_EMPTY_INPUT


def test_distance_known_answers(oracle):
    # src/core/distance.rs:130-165
    a, b = [1, 2, 3], [4, 6, 8]
    assert oracle.distance(O.SQUARED_EUCLIDEAN, a, b) == F(50.0)
    assert oracle.distance(O.EUCLIDEAN, a, b) == np.sqrt(F(50.0))
    assert oracle.distance(O.MANHATTAN, a, b) == F(12.0)
    assert abs(oracle.distance(O.COSINE, [1, 0], [0, 1]) - 1.0) < 1e-6
    assert abs(oracle.distance(O.COSINE, [1, 1], [1, 1]) - 0.0) < 1e-6


def test_distance_known_answers_pyvq(oracle):
    # pyvq/tests/test_distance.py:30-59
    a, b = [1.0, 2.0], [3.0, 4.0]
    assert np.isclose(oracle.distance(O.EUCLIDEAN, a, b), 2.8284, rtol=1e-4)
    assert np.isclose(oracle.distance(O.SQUARED_EUCLIDEAN, a, b), 8.0, rtol=1e-4)
    assert np.isclose(oracle.distance(O.COSINE, a, b), 0.01613, rtol=1e-3)
    assert np.isclose(oracle.distance(O.MANHATTAN, a, b), 4.0, rtol=1e-4)


def test_distance_dimension_mismatch(oracle):
    # src/core/distance.rs:167-173
    with pytest.raises(O.OracleError) as e:
        oracle.distance(O.EUCLIDEAN, [1, 2], [1])
    assert e.value.code == O.ERR_DIMENSION_MISMATCH


def test_cosine_edge_cases(oracle):
    # tests/regression_tests.rs:241-275 (scalar path: exactly 1.0; clamp into [0,1])
    assert oracle.distance(O.COSINE, [0, 0, 0], [1, 2, 3]) == F(1.0)
    assert oracle.distance(O.COSINE, [1e-20, 1e-20, 1e-20], [1, 2, 3]) == F(1.0)
    d = oracle.distance(O.COSINE, [1, 0, 0], [1, 0, 0])
    assert 0.0 <= d <= 1.0 and abs(d) < 1e-6
    # tests/integration_tests.rs:646-670
    assert oracle.distance(O.COSINE, [0, 0, 0], [0, 0, 0]) == F(1.0)
    # opposite vectors: 1 - (-1) = 2 clamps to 1 (src/core/distance.rs:118)
    assert oracle.distance(O.COSINE, [1, 2], [-1, -2]) == F(1.0)


def test_nan_and_infinity_propagation(oracle):
    # tests/integration_tests.rs:606-643
    a, b = [1.0, np.nan, 3.0], [1.0, 2.0, 3.0]
    for metric in (O.EUCLIDEAN, O.MANHATTAN, O.SQUARED_EUCLIDEAN):
        assert math.isnan(oracle.distance(metric, a, b))
    for metric in (O.EUCLIDEAN, O.MANHATTAN):
        r = oracle.distance(metric, [np.inf, 0.0], [0.0, 0.0])
        assert math.isinf(r) and r > 0
        assert math.isinf(oracle.distance(metric, [np.inf], [-np.inf]))


def test_pq_single_training_vector_k1(oracle):
    # tests/integration_tests.rs:324-332: N = 1, k = 1, m = 2 -> the only row is every
    # centroid whatever the RNG does; quantize returns f16 of the row itself.
    rows = np.array([[1, 2, 3, 4]], F)
    cb, iters = oracle.pq_fit(rows, m=2, k=1, max_iters=10, init_rows=[[0], [0]])
    np.testing.assert_array_equal(cb.reshape(-1), rows.reshape(-1))
    codes, f16 = oracle.pq_encode(O.EUCLIDEAN, rows, cb)
    assert codes.tolist() == [[0, 0]]
    np.testing.assert_array_equal(f16.view(np.float16), rows.astype(np.float16))


@pytest.mark.parametrize("perm", [[0, 1], [1, 0]])
def test_pq_k_equals_n_distinct_rows(oracle, perm):
    # tests/regression_tests.rs:357-363: N = 2, k = 2 with distinct rows: each row is its
    # own centroid for either sampling order, so quantize(x_i) == f16(x_i).
    rows = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], F)
    cb, _ = oracle.pq_fit(rows, m=2, k=2, max_iters=10, init_rows=[perm, perm])
    _, f16 = oracle.pq_encode(O.MANHATTAN, rows, cb)
    np.testing.assert_array_equal(f16.view(np.float16), rows.astype(np.float16))


def test_lbg_epsilon_convergence_case(oracle):
    # tests/regression_tests.rs:208-225: 4 near-duplicate points, k = 2, max_iters = 100.
    data = np.array([[1.0, 1.0], [1.0001, 1.0001], [10.0, 10.0], [10.0001, 10.0001]], F)
    for init in ([0, 2], [0, 1], [2, 3], [1, 3], [3, 0]):
        cent, iters, used = oracle.lloyd(data, 2, 100, init)
        assert cent.shape == (2, 2)
        assert iters <= 3  # "should converge quickly with epsilon comparison"
        got = sorted(cent[:, 0].tolist())
        assert abs(got[0] - 1.00005) < 1e-3 and abs(got[1] - 10.00005) < 1e-3


def test_lbg_parameter_errors(oracle):
    # src/core/vector.rs:396-410 / 555-575
    data = np.ones((3, 2), F)
    with pytest.raises(O.OracleError) as e:
        oracle.lloyd(data, 0, 10, [])
    assert e.value.code == O.ERR_INVALID_PARAMETER
    with pytest.raises(O.OracleError) as e:
        oracle.lloyd(data, 4, 10, [0, 1, 2, 0])
    assert e.value.code == O.ERR_INVALID_PARAMETER
    with pytest.raises(O.OracleError) as e:
        oracle.lloyd(np.zeros((0, 2), F), 1, 10, [0])
    assert e.value.code == O.ERR_EMPTY_INPUT


def test_lbg_max_iters_zero_returns_init_rows(oracle):
    # src/core/vector.rs:415, 460
    data = np.arange(20, dtype=F).reshape(10, 2)
    cent, iters, _ = oracle.lloyd(data, 3, 0, [7, 2, 5])
    assert iters == 0
    np.testing.assert_array_equal(cent, data[[7, 2, 5]])


def test_pq_validation_errors(oracle):
    # src/pq.rs:91-117, tests/integration_tests.rs:134-203
    rows = np.ones((4, 6), F)
    with pytest.raises(O.OracleError) as e:
        oracle.pq_fit(np.zeros((0, 6), F), 2, 1, 1, [[0], [0]])
    assert e.value.code == O.ERR_EMPTY_INPUT
    with pytest.raises(O.OracleError) as e:  # dim < m
        oracle.pq_fit(rows, 8, 1, 1, [[0]] * 8)
    assert e.value.code == O.ERR_INVALID_PARAMETER
    with pytest.raises(O.OracleError) as e:  # dim % m != 0
        oracle.pq_fit(rows, 4, 1, 1, [[0]] * 4)
    assert e.value.code == O.ERR_INVALID_PARAMETER


def test_tsvq_identical_vectors_root_is_leaf(oracle):
    # src/tsvq.rs:273-284, tests/integration_tests.rs:335-343: every value equals the median,
    # so `left` takes all rows and both children are None (src/tsvq.rs:88-108).
    vec = np.array([1, 2, 3, 4, 5], F)
    rows = np.tile(vec, (10, 1))
    tree = oracle.tsvq_build(rows, 3)
    assert tree["centroids"].shape == (1, 5)
    assert tree["left"].tolist() == [-1] and tree["right"].tolist() == [-1]
    leaf, f16 = oracle.tsvq_encode(O.SQUARED_EUCLIDEAN, vec[None, :], tree)
    assert leaf.tolist() == [0]
    assert np.all(np.abs(f16.view(np.float16).astype(F) - vec) < 1e-2)


def test_tsvq_two_rows(oracle):
    # tests/regression_tests.rs:367-373: N = 2, depth 2 -> root + two single-row leaves
    rows = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], F)
    tree = oracle.tsvq_build(rows, 2)
    assert tree["centroids"].shape[0] == 3
    np.testing.assert_array_equal(tree["centroids"][0], [3, 4, 5, 6])
    np.testing.assert_array_equal(tree["centroids"][tree["left"][0]], rows[0])
    np.testing.assert_array_equal(tree["centroids"][tree["right"][0]], rows[1])
    leaf, f16 = oracle.tsvq_encode(O.COSINE, rows, tree)
    # cosine distance picks by angle; both rows map to a leaf and come back as a leaf mean
    assert set(leaf.tolist()) <= {1, 2}


def test_tsvq_nan_row_is_tolerated(oracle):
    # tests/regression_tests.rs:282-297 (partial NaN: must not panic)
    rows = np.array([[1, 2, 3, 4], [5, np.nan, 7, 8], [9, 10, 11, 12]], F)
    tree = oracle.tsvq_build(rows, 2)
    assert tree["centroids"].shape[0] >= 1


def test_tsvq_all_nan_split_column_is_reference_panic(oracle):
    # src/tsvq.rs:77-78 with an empty `values`: the reference panics; the oracle says so.
    rows = np.full((3, 1), np.nan, F)
    with pytest.raises(O.OracleError) as e:
        oracle.tsvq_build(rows, 2)
    assert e.value.code == O.ERR_REFERENCE_PANICS


def test_tsvq_structured_dataset_builds(oracle):
    # src/tsvq.rs:286-299 and tests/regression_tests.rs:381-392 (construction + encode succeed)
    data = np.array([[(i + j) % 50 for j in range(10)] for i in range(100)], F)
    tree = oracle.tsvq_build(data, 3)
    assert 1 <= tree["centroids"].shape[0] <= 15
    leaf, f16 = oracle.tsvq_encode(O.SQUARED_EUCLIDEAN, data, tree)
    assert f16.shape == (100, 10)
    is_leaf = (tree["left"] < 0) & (tree["right"] < 0)
    assert np.all(is_leaf[leaf])
    big = np.array([[(i + j) % 100 for j in range(32)] for i in range(1000)], F)
    t2 = oracle.tsvq_build(big, 5)
    assert int(t2["node_rows"][0]) == 1000


def test_encode_is_deterministic(oracle):
    # tests/integration_tests.rs:40-53, tests/property_tests.rs:196-206
    rng = np.random.default_rng(0)
    rows = rng.random((50, 8), dtype=F)
    cb, _ = oracle.pq_fit(rows, 2, 4, 5, [[0, 1, 2, 3]] * 2, reseed_rows=[[0] * 16] * 2)
    a = oracle.pq_encode(O.EUCLIDEAN, rows, cb)
    b = oracle.pq_encode(O.EUCLIDEAN, rows, cb)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_f16_conversion_matches_ieee_rne(oracle):
    # half::f16::from_f32 / to_f32 are IEEE conversions (src/pq.rs:194, 208); numpy's
    # float16 cast is the same rounding.
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.standard_normal(2000).astype(F),
        (rng.standard_normal(2000) * 1e-6).astype(F),  # f16 subnormal range
        (rng.standard_normal(500) * 7e4).astype(F),  # overflow boundary
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8,
                  2.9802326e-8, 6.1035156e-5, 6.0975552e-5, np.inf, -np.inf, 1.0009766, 1.00048828125,
                  1.0014648], F),
    ])
    got = oracle.f32_to_f16_bits(vals)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    np.testing.assert_array_equal(got, want)
    # NaN stays NaN
    assert np.isnan(np.array(oracle.f32_to_f16_bits([np.nan]), np.uint16).view(np.float16)[0])
    # every f16 bit pattern converts back exactly
    allbits = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    back = oracle.f16_bits_to_f32(allbits)
    ref = allbits.view(np.float16).astype(F)
    ok = (back == ref) | (np.isnan(back) & np.isnan(ref))
    assert ok.all()
    # and round-trips through from_f32
    finite = np.isfinite(ref)
    np.testing.assert_array_equal(oracle.f32_to_f16_bits(ref[finite]), allbits[finite])
