"""The host mirror (`vq_amd`) behaves like the reference's Python surface (`pyvq`): these tests
restate pyvq/tests/test_pq.py, test_tsvq.py, test_distance.py and the RNG-independent parts of
test_integrations.py / test_regressions.py with `vq_amd` in place of `pyvq` (citations per test).
Validation errors are raised before anything touches the device, so those run on CPU; the rest
needs a GPU (`-m gpu`).
"""
import numpy as np
import pytest

import vq_amd as pyvq

F = np.float32


# ---------------------------------------------------------------- CPU: validation + surface ----
def test_distance_constructors_and_repr():
    # pyvq/tests/test_distance.py:7-27, 73-100
    for name in ("euclidean", "squared_euclidean", "manhattan", "cosine"):
        d = pyvq.Distance(name)
        assert repr(d) == f"Distance('{name}')"
        assert d == getattr(pyvq.Distance, name)()
    assert pyvq.Distance("SquaredEuclidean") == pyvq.Distance.squared_euclidean()
    assert pyvq.Distance("cosine_distance") == pyvq.Distance.cosine()
    with pytest.raises(ValueError, match="Invalid distance metric"):
        pyvq.Distance("chebyshev")
    # Distance::name, tests/regression_tests.rs:347-352
    assert pyvq.Distance.cosine().name() == "cosine"
    assert pyvq.Distance.squared_euclidean().name() == "squared_euclidean"


def test_distance_length_mismatch_is_value_error():
    # pyvq/tests/test_distance.py:62-70; message of VqError::DimensionMismatch
    with pytest.raises(ValueError, match="Dimension mismatch: expected 2, found 3"):
        pyvq.Distance.euclidean().compute(np.zeros(2, F), np.zeros(3, F))


def test_pq_empty_training_raises():
    # pyvq/tests/test_pq.py:78-82 ("empty"), pyvq/src/pq.rs:60-62
    with pytest.raises(ValueError, match="empty"):
        pyvq.ProductQuantizer(np.array([]).reshape(0, 8).astype(F), 2, 4)
    # Rust surface: EmptyInput display text, tests/integration_tests.rs:134-141
    with pytest.raises(pyvq.EmptyInput, match="Empty input: at least one vector is required"):
        pyvq.ProductQuantizer([], 2, 4)


def test_pq_invalid_subspaces_and_centroids():
    # pyvq/tests/test_pq.py:85-90; src/pq.rs:106-117; src/core/vector.rs:399-410
    with pytest.raises(pyvq.InvalidParameter, match=r"Invalid parameter 'm': dimension \(7\) must be divisible by m"):
        pyvq.ProductQuantizer(np.random.rand(50, 7).astype(F), 2, 4)
    with pytest.raises(pyvq.InvalidParameter, match=r"Invalid parameter 'm': must be at most the data dimension \(4\)"):
        pyvq.ProductQuantizer(np.random.rand(50, 4).astype(F), 8, 4)
    with pytest.raises(pyvq.InvalidParameter, match=r"Invalid parameter 'k': must be greater than 0"):
        pyvq.ProductQuantizer(np.random.rand(50, 8).astype(F), 2, 0)
    with pytest.raises(pyvq.InvalidParameter, match=r"not enough data points \(5\) for 8 clusters"):
        pyvq.ProductQuantizer(np.random.rand(5, 8).astype(F), 2, 8)
    assert issubclass(pyvq.InvalidParameter, ValueError)


def test_pq_ragged_rows_dimension_mismatch():
    # tests/regression_tests.rs:69-88 (DimensionMismatch { expected, found })
    with pytest.raises(pyvq.DimensionMismatch, match="expected 4, found 3"):
        pyvq.ProductQuantizer([[1, 2, 3, 4], [1, 2, 3]], 2, 1)
    with pytest.raises(pyvq.DimensionMismatch):
        pyvq.TSVQ([[1, 2, 3, 4], [1, 2, 3]], 2)


def test_tsvq_empty_training_raises():
    # pyvq/tests/test_tsvq.py:60-64, src/tsvq.rs:196-198
    with pytest.raises(ValueError, match="empty"):
        pyvq.TSVQ(np.array([]).reshape(0, 8).astype(F), 3)


def test_host_rng_is_deterministic_and_distinct():
    from vq_amd.rng import HostRng

    a, b = HostRng(42), HostRng(42)
    ra, rb = a.choose_multiple(1000, 256), b.choose_multiple(1000, 256)
    assert ra == rb and len(set(ra)) == 256 and all(0 <= r < 1000 for r in ra)
    assert HostRng(43).choose_multiple(1000, 256) != ra
    assert sorted(HostRng(1).choose_multiple(7, 7)) == list(range(7))
    assert all(0 <= HostRng(s).choose(10) < 10 for s in range(50))


# ------------------------------------------------------------------------------- GPU ----
@pytest.mark.gpu
def test_product_quantizer_surface():
    # pyvq/tests/test_pq.py:7-75
    training = np.random.default_rng(0).random((100, 16), dtype=F)
    pq = pyvq.ProductQuantizer(training_data=training, num_subspaces=4, num_centroids=8, max_iters=10, seed=42)
    assert (pq.dim, pq.num_subspaces, pq.sub_dim) == (16, 4, 4)
    assert repr(pq) == "ProductQuantizer(dim=16, num_subspaces=4, sub_dim=4)"
    assert pq.distance_metric() == "euclidean"  # default metric, pyvq/src/pq.rs:73-75
    codes = pq.quantize(training[0].copy())
    assert isinstance(codes, np.ndarray) and codes.dtype == np.float16 and len(codes) == 16
    rec = pq.dequantize(codes)
    assert rec.dtype == np.float32 and len(rec) == 16
    np.testing.assert_array_equal(rec, codes.astype(F))
    with pytest.raises(ValueError, match="Dimension mismatch: expected 16, found 10"):
        pq.quantize(np.zeros(10, F))
    with pytest.raises(ValueError, match="Dimension mismatch"):
        pq.dequantize(np.zeros(3, np.float16))


@pytest.mark.gpu
def test_quantize_is_deterministic_and_seeded():
    # tests/integration_tests.rs:40-53; pyvq/tests/test_properties.py determinism
    X = np.random.default_rng(1).random((300, 8), dtype=F)
    a = pyvq.ProductQuantizer(X, 2, 16, seed=7)
    b = pyvq.ProductQuantizer(X, 2, 16, seed=7)
    np.testing.assert_array_equal(a.codebooks, b.codebooks)
    np.testing.assert_array_equal(a.quantize(X[3]), a.quantize(X[3]))
    np.testing.assert_array_equal(a.quantize_batch(X)[3], a.quantize(X[3]))


@pytest.mark.gpu
def test_all_metrics_construct_and_encode():
    # tests/integration_tests.rs:246-264
    X = np.random.default_rng(2).random((200, 12), dtype=F)
    for d in (pyvq.Distance.euclidean(), pyvq.Distance.squared_euclidean(), pyvq.Distance.manhattan(),
              pyvq.Distance.cosine()):
        pq = pyvq.ProductQuantizer(X, 3, 8, distance=d)
        assert pq.quantize(X[0]).shape == (12,)
        assert pq.distance_metric() == d.name()
        t = pyvq.TSVQ(X, 3, d)
        assert t.quantize(X[0]).shape == (12,) and t.distance_metric() == d.name()


@pytest.mark.gpu
def test_single_training_vector_k1_and_k_equals_n():
    # tests/integration_tests.rs:324-332: the value is forced whatever the RNG does
    row = np.array([[1, 2, 3, 4]], F)
    pq = pyvq.ProductQuantizer(row, 2, 1, 10, pyvq.Distance.euclidean(), 42)
    np.testing.assert_array_equal(pq.quantize(row[0]), row[0].astype(np.float16))
    # tests/regression_tests.rs:357-363: k = N distinct rows
    X = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], F)
    pq = pyvq.ProductQuantizer(X, 2, 2, 10, pyvq.Distance.manhattan(), 42)
    for r in X:
        np.testing.assert_array_equal(pq.quantize(r), r.astype(np.float16))


@pytest.mark.gpu
def test_pq_reconstruction_quality_bounds():
    # pyvq/tests/test_integrations.py:42-62 (RMSE < 2.0) and 175-197 (10k x 64, m=8, k=256)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((200, 16)).astype(F)
    pq = pyvq.ProductQuantizer(X, 4, 16, max_iters=20)
    rec = pq.quantize_batch(X).astype(F)
    assert np.sqrt(((X - rec) ** 2).mean()) < 2.0
    X = rng.random((10_000, 64), dtype=F)
    pq = pyvq.ProductQuantizer(X, 8, 256, max_iters=10)
    codes = pq.encode(X)
    assert codes.shape == (10_000, 8) and codes.dtype == np.uint8
    rec = pq.decode(codes)
    assert ((X - rec) ** 2).mean() < 0.02
    np.testing.assert_array_equal(pq.quantize_batch(X[:50]), rec[:50].astype(np.float16))


@pytest.mark.gpu
def test_distance_known_answers():
    # pyvq/tests/test_distance.py:30-59, tests/regression_tests.rs:241-261
    a, b = np.array([1.0, 2.0], F), np.array([3.0, 4.0], F)
    assert np.isclose(pyvq.Distance.euclidean().compute(a, b), 2.8284, rtol=1e-4)
    assert np.isclose(pyvq.Distance.squared_euclidean().compute(a, b), 8.0, rtol=1e-4)
    assert np.isclose(pyvq.Distance.cosine().compute(a, b), 0.01613, rtol=1e-3)
    assert np.isclose(pyvq.Distance.manhattan().compute(a, b), 4.0, rtol=1e-4)
    assert pyvq.Distance.cosine().compute(np.zeros(3, F), np.array([1, 2, 3], F)) == 1.0
    assert pyvq.Distance.cosine().compute(np.full(3, 1e-20, F), np.array([1, 2, 3], F)) == 1.0
    assert np.isnan(pyvq.Distance.euclidean().compute(np.array([1, np.nan, 3], F), np.array([1, 2, 3], F)))


@pytest.mark.gpu
def test_tsvq_surface_and_quality():
    # pyvq/tests/test_tsvq.py:7-57, 68-86
    rng = np.random.default_rng(4)
    X = rng.random((100, 8), dtype=F)
    t = pyvq.TSVQ(training_data=X, max_depth=3)
    assert t.dim == 8 and repr(t) == "TSVQ(dim=8)"
    q = t.quantize(X[0])
    assert q.dtype == np.float16 and len(q) == 8
    assert t.dequantize(q).dtype == np.float32
    with pytest.raises(ValueError, match="Dimension mismatch"):
        t.quantize(np.zeros(10, F))
    training = np.vstack([rng.standard_normal((50, 4)), rng.standard_normal((50, 4)) + 10]).astype(F)
    t = pyvq.TSVQ(training, max_depth=3)
    rec = t.dequantize(t.quantize(training[0]))
    assert np.linalg.norm(rec) < np.linalg.norm(rec - 10)


@pytest.mark.gpu
def test_get_simd_backend_names_the_device_backend():
    assert "gfx950" in pyvq.get_simd_backend()


@pytest.mark.gpu
def test_code_index_reproduces_quantize_output(tmp_path):
    # m code bytes per vector + codebooks carry what quantize() returns (src/pq.rs:183-196)
    from vq_amd.store import PQIndex

    X = np.random.default_rng(8).random((3000, 32), dtype=F)
    pq = pyvq.ProductQuantizer(X, 8, 64, max_iters=5, distance=pyvq.Distance.squared_euclidean())
    idx = PQIndex.from_quantizer(pq, X)
    np.testing.assert_array_equal(idx.reconstruct_f16().view(np.uint16), pq.quantize_batch(X).view(np.uint16))
    np.testing.assert_array_equal(idx.reconstruct(), pq.decode(idx.codes))
    idx.save(tmp_path / "i.vqpq")
    back = PQIndex.load(tmp_path / "i.vqpq")
    pq2 = pyvq.ProductQuantizer.from_codebooks(back.codebooks, back.distance)
    np.testing.assert_array_equal(pq2.encode(X[:100]), idx.codes[:100])
    assert 4 * 32 * len(idx) / idx.codes.nbytes == 16.0  # vs 2x for the f16 form


@pytest.mark.gpu
def test_eval_report_has_the_reference_lines(capsys):
    # src/bin/eval_pq.rs:33-69, src/bin/eval_tsvq.rs:27-56
    from vq_amd import evalcli

    assert evalcli.main(["pq", "--samples", "1000", "2000", "--dim", "32", "--m", "4", "--k", "16"]) == 0
    out = capsys.readouterr().out
    assert out.startswith("Product Quantizer Evaluation\n============================\n")
    assert out.count("Samples: ") == 2 and "Samples: 2000" in out
    for key in ("Training time:", "Quantization time:", "Reconstruction error:", "Recall@10:"):
        assert out.count(key) == 2
    err = float(out.split("Reconstruction error: ")[1].split()[0])
    assert 0.0 < err < 1.0 / 12  # better than quantizing Uniform[0,1) to its mean
    assert evalcli.main(["tsvq", "--samples", "1000", "--dim", "32", "--max-depth", "4", "--json"]) == 0
    import json

    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    res = json.loads(lines[0])
    assert set(res) >= {"n_samples", "n_dims", "training_time_ms", "quantization_time_ms", "reconstruction_error",
                        "recall", "memory_reduction_ratio"}
