"""Opt-in `simd`-compatible cosine (VQHIP_COSINE_UNCLAMPED = 4): 1 - dot / (|a| |b|) with no EPSILON rule and no clamp
(the reference's simd build, src/core/distance.rs:97-105; hsdlib's summation order is unknown, so this id is UNPINNED:
the oracle states the semantics, the GPU paths must equal the oracle)."""
import numpy as np
import pytest

import oracle as O
from vq_amd import Distance

F = np.float32


def test_oracle_semantics_differ_from_the_clamped_metric_only_where_the_source_says(oracle):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((200, 16)).astype(F)
    b = rng.standard_normal((200, 16)).astype(F)
    d3 = np.array([oracle.distance(O.COSINE, x, y) for x, y in zip(a, b)], F)
    d4 = np.array([oracle.distance(O.COSINE_UNCLAMPED, x, y) for x, y in zip(a, b)], F)
    inside = d4 <= 1.0
    np.testing.assert_array_equal(d3[inside], d4[inside])  # same three sums, same quotient
    assert (d4[~inside] > 1.0).all() and (d3[~inside] == 1.0).all()  # negative cosines: the clamp is gone
    assert (~inside).sum() > 50
    z = np.zeros(16, F)
    assert oracle.distance(O.COSINE, z, a[0]) == 1.0  # EPSILON rule
    assert np.isnan(oracle.distance(O.COSINE_UNCLAMPED, z, a[0]))  # 0 / 0
    assert Distance("cosine_simd").metric == 4 and Distance.cosine_unclamped().name() == "cosine_unclamped"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3000, 32, 4, 64), (2000, 24, 3, 256), (1500, 100, 10, 17), (800, 7, 1, 5)])
def test_pq_encode_equals_the_oracle(oracle, shape):
    from vq_amd import _lib

    n, d, m, k = shape
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d)).astype(F)
    X[3] = 0  # zero norm: every distance NaN -> centroid 0 (the `<` scan never replaces the first)
    X[4] *= F(1e-30)
    cb = rng.standard_normal((m, k, d // m)).astype(F)
    cb[0, 1] = 0
    enc = _lib.PQEncoder(cb, _lib.COSINE_UNCLAMPED)
    codes, f16 = enc.encode(X, True, True)
    want_codes, want_f16 = oracle.pq_encode(O.COSINE_UNCLAMPED, X, cb, threads=0)
    np.testing.assert_array_equal(codes, want_codes)
    np.testing.assert_array_equal(np.asarray(f16).view(np.uint16), np.asarray(want_f16).view(np.uint16))
    # and it really is another metric: some rows decide differently under the clamp
    c3, _ = oracle.pq_encode(O.COSINE, X, cb, want_f16=False, threads=0)
    assert (np.asarray(c3) != np.asarray(want_codes)).any()
    one = enc.encode(X[5:6], True, False)[0]
    np.testing.assert_array_equal(one, want_codes[5:6])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4000, 24, 7), (3000, 128, 6), (2000, 384, 4), (2500, 10, 6)])
def test_tsvq_descent_equals_the_oracle(oracle, shape):
    from vq_amd import TSVQ

    n, d, depth = shape
    rng = np.random.default_rng(6)
    X = rng.standard_normal((n, d)).astype(F)
    Q = np.concatenate([rng.standard_normal((3000, d)).astype(F), X[:300], np.zeros((2, d), F)])
    tree = oracle.tsvq_build(X, depth)
    t = TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], Distance("cosine_unclamped"))
    want_leaf, want_f16 = oracle.tsvq_encode(O.COSINE_UNCLAMPED, Q, tree, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    assert not t.last_encode_stats()[0]  # exact walk only: no screen is proven for this id
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)
    np.testing.assert_array_equal(t.quantize(Q[7]).view(np.uint16), want_f16[7])
    clamped_leaf, _ = oracle.tsvq_encode(O.COSINE, Q, tree, want_f16=False, threads=0)
    assert (clamped_leaf != want_leaf).any()
