"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py).

CPU half: the oracle still reproduces every fixture bit for bit.
GPU half (-m gpu): the HIP path through the C ABI reproduces them -- codes, counts and f16
outputs exactly, centroids within the stated tolerance -- without needing the oracle to be
rebuilt identically on the GPU box.
"""
import os

import numpy as np
import pytest

import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
METRICS = ((0, "sqeuclid"), (1, "euclid"), (2, "manhattan"), (3, "cosine"))
ENCODE_FILES = ["encode_uniform_m4_k16.npz", "encode_normal_m8_k256.npz", "encode_adversarial_m4_k32.npz"]
F = np.float32


def _load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("name", ENCODE_FILES)
def test_oracle_reproduces_encode_fixture(oracle, name):
    g = _load(name)
    for metric, mname in METRICS:
        codes, f16 = oracle.pq_encode(metric, g["X"], g["codebooks"])
        np.testing.assert_array_equal(codes.astype(np.uint8), g[f"codes_{mname}"])
        np.testing.assert_array_equal(f16, g[f"f16_{mname}"])


def test_oracle_reproduces_lloyd_fixtures(oracle):
    g = _load("lloyd_step_m2_k16.npz")
    X, init = g["X"], g["init_rows"]
    for s in range(2):
        c0 = X[init[s].astype(np.int64), s * 16:(s + 1) * 16]
        c1, assign, counts, changed = oracle.lloyd_step(X[:, s * 16:(s + 1) * 16], c0)
        assert c1.tobytes() == g[f"centroids_out_{s}"].tobytes()
        np.testing.assert_array_equal(assign.astype(np.uint8), g[f"assign_{s}"])
        np.testing.assert_array_equal(counts, g[f"counts_{s}"])
        assert changed == bool(g[f"changed_{s}"])
    g = _load("pq_fit_m2_k16.npz")
    cb, iters = oracle.pq_fit(g["X"], 2, 16, 10, g["init_rows"], reseed_rows=g["reseed_rows"])
    assert cb.tobytes() == g["codebooks"].tobytes()
    np.testing.assert_array_equal(iters, g["iters"])


def test_oracle_reproduces_tsvq_fixture(oracle):
    g = _load("tsvq_depth5.npz")
    tree = oracle.tsvq_build(g["X"], 5)
    assert tree["centroids"].tobytes() == g["centroids"].tobytes()
    np.testing.assert_array_equal(tree["left"], g["left"])
    np.testing.assert_array_equal(tree["right"], g["right"])
    for metric, mname in METRICS:
        leaf, f16 = oracle.tsvq_encode(metric, g["Q"], tree)
        np.testing.assert_array_equal(leaf, g[f"leaf_{mname}"])
        np.testing.assert_array_equal(f16, g[f"f16_{mname}"])


# ------------------------------------------------------------------------------- GPU ----
@pytest.mark.gpu
@pytest.mark.parametrize("name", ENCODE_FILES)
def test_gpu_reproduces_encode_fixture(name):
    from vq_amd import _lib

    g = _load(name)
    for metric, mname in METRICS:
        engines = [_lib.ENGINE_AUTO, _lib.ENGINE_EXACT]
        if metric in (0, 1):
            engines += [_lib.ENGINE_MFMA, _lib.ENGINE_MFMA_BF16]
        for engine in engines:
            enc = _lib.PQEncoder(g["codebooks"], metric)
            enc.set_engine(engine)
            codes, f16 = enc.encode(g["X"])
            np.testing.assert_array_equal(codes, g[f"codes_{mname}"])
            np.testing.assert_array_equal(f16.view(np.uint16), g[f"f16_{mname}"])
            enc.close()


@pytest.mark.gpu
def test_gpu_reproduces_lloyd_step_fixture():
    from vq_amd import _lib

    g = _load("lloyd_step_m2_k16.npz")
    ds = _lib.Dataset.from_host(g["X"])
    km = _lib.KMeans(ds, 2, 16)
    km.init_from_rows(g["init_rows"])
    counts, changed = km.step()
    assign, cent = km.get_assignments(), km.get_centroids()
    for s in range(2):
        np.testing.assert_array_equal(assign[:, s], g[f"assign_{s}"])
        np.testing.assert_array_equal(counts[s], g[f"counts_{s}"])
        ref = g[f"centroids_out_{s}"]
        assert np.max(np.abs(cent[s] - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-5
        assert bool(changed[s]) == bool(g[f"changed_{s}"])
    km.close()
    ds.close()


@pytest.mark.gpu
def test_gpu_full_fit_fixture_quality():
    from vq_amd import _lib
    from vq_amd.pq import fit_codebooks

    g = _load("pq_fit_m2_k16.npz")
    ds = _lib.Dataset.from_host(g["X"])
    stats = {}
    cb = fit_codebooks(ds, 2, 16, 10, init_rows=g["init_rows"], reseed_rows=g["reseed_rows"], stats=stats)
    ds.close()
    ref = g["codebooks"]
    X = g["X"]

    def mse(c):
        enc = _lib.PQEncoder(c, _lib.SQUARED_EUCLIDEAN)
        codes, _ = enc.encode(X, want_f16=False)
        rec = enc.decode(codes)
        enc.close()
        return float(((X - rec) ** 2).mean())

    assert abs(mse(cb) - mse(ref)) / mse(ref) < 1e-3
