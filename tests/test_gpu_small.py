"""The per-vector latency path (<= 8 host rows per call: one kernel over mapped pinned memory) is the
reference's `quantize(&[f32])` shape (src/pq.rs:167-199, src/tsvq.rs:239-255): same bits as the oracle
and as the batch pipeline, for every metric."""
import numpy as np
import pytest

import oracle as O
from vq_amd import TSVQ, Distance, _lib

import os

pytestmark = pytest.mark.gpu
F = np.float32
SCALE = int(os.environ.get("VQ_FUZZ_SCALE", "1"))
NAMES = ["squared_euclidean", "euclidean", "manhattan", "cosine"]


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_small_pq_encode_matches_oracle_and_batch_path(oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    sd = int(rng.choice([1, 3, 8, 16, 24, 40, 100]))
    m = int(rng.choice([1, 2, 8, 16]))
    k = int(rng.choice([1, 7, 64, 200, 256]))
    metric = int(rng.integers(0, 4))
    kind = rng.choice(["normal", "lattice", "zeros"])
    d = sd * m
    cb = (rng.integers(-2, 3, (m, k, sd)) if kind == "lattice" else rng.standard_normal((m, k, sd))).astype(F)
    if k > 2:
        cb[:, k - 1] = cb[:, 0]
    enc = _lib.PQEncoder(cb, metric)
    for n in (1, 3, 8):
        X = (rng.integers(-2, 3, (n, d)) if kind == "lattice" else rng.standard_normal((n, d))).astype(F)
        if kind == "zeros":
            X[0] = 0
        if seed % 5 == 0:
            X[-1, 0] = np.nan
        codes, f16 = enc.encode(X)  # n <= 8: latency path
        want_c, want_f = oracle.pq_encode(metric, X, cb, threads=1)
        np.testing.assert_array_equal(codes.astype(np.uint32), want_c, err_msg=f"sd={sd} m={m} k={k} metric={metric} {kind} n={n}")
        same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
        assert same.all()
        big = np.concatenate([X, np.zeros((9, d), F)])  # 9+ rows: batch pipeline
        codes_b, f16_b = enc.encode(big)
        np.testing.assert_array_equal(codes_b[:n], codes)
        np.testing.assert_array_equal(f16_b[:n].view(np.uint16), f16.view(np.uint16))
    enc.close()


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_small_tsvq_encode_matches_oracle(oracle, seed):
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.choice([1, 5, 24, 64, 128, 130, 300]))
    metric = int(rng.integers(0, 4))
    X = rng.standard_normal((1500, d)).astype(F)
    if seed % 3 == 0:
        X = rng.integers(0, 3, (1500, d)).astype(F)
    tree = oracle.tsvq_build(X, int(rng.integers(1, 8)))
    t = TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], Distance(NAMES[metric]))
    for n in (1, 2, 8):
        Q = np.concatenate([rng.standard_normal((n - 1, d)).astype(F), X[:1]]) if n > 1 else X[5:6]
        want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, tree, threads=1)
        np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf, err_msg=f"d={d} metric={metric} n={n}")
        np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)
        assert not t.last_encode_stats()[0]
    np.testing.assert_array_equal(t.quantize(X[7]).view(np.uint16), oracle.tsvq_encode(metric, X[7:8], tree, threads=1)[1][0])
