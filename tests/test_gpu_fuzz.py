"""Randomised parity sweep (GPU vs oracle): shapes, metrics, engines and data kinds drawn from a
fixed seed -- the cases nobody thought of.  Every draw must be bit-exact (codes, f16, leaves, tree)."""
import numpy as np
import pytest

import oracle as O
from vq_amd import TSVQ, Distance, _lib
from vq_amd.tsvq import build_tree

import os

pytestmark = pytest.mark.gpu
F = np.float32
SCALE = int(os.environ.get("VQ_FUZZ_SCALE", "1"))  # VQ_FUZZ_SCALE=20 for a long hunt
KINDS = ["uniform", "normal", "lattice", "clustered", "offset", "mixed_scale", "sparse", "dupes"]


def _draw_data(rng, n, d, kind):
    if kind == "uniform":
        return rng.random((n, d), dtype=F)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(F)
    if kind == "lattice":
        return rng.integers(-2, 3, (n, d)).astype(F)
    if kind == "clustered":
        c = rng.standard_normal((17, d)).astype(F) * 3
        return (c[rng.integers(0, 17, n)] + 0.01 * rng.standard_normal((n, d))).astype(F)
    if kind == "offset":
        return (rng.random((n, d), dtype=F) * F(0.5) + F(rng.choice([-200.0, 33.0, 4096.0]))).astype(F)
    if kind == "mixed_scale":  # columns spanning 12 orders of magnitude
        return (rng.standard_normal((n, d)) * np.exp(rng.uniform(-14, 14, d))).astype(F)
    if kind == "sparse":
        x = rng.standard_normal((n, d)).astype(F)
        x[rng.random((n, d)) < 0.8] = 0
        return x
    if kind == "dupes":  # many identical rows
        base = rng.standard_normal((max(3, n // 50), d)).astype(F)
        return base[rng.integers(0, len(base), n)]
    raise ValueError(kind)


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_fuzz_pq_encode(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    sd = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10, 10, 11, 12, 14, 16, 16, 18, 20, 22, 24, 25, 28, 30, 32, 32, 33, 36, 40, 48, 48, 50, 57, 60, 64, 64, 68, 72, 80, 88, 96, 100, 112, 120, 128, 128, 130, 140, 176]))
    m = int(rng.choice([1, 2, 3, 4, 8, 16]))
    d = sd * m
    k = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 65, 100, 100, 128, 128, 200, 225, 240, 255, 256, 256]))
    n = int(rng.integers(1, 3000))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    metric = int(rng.integers(0, 4))
    X = _draw_data(rng, n, d, kind)
    if rng.random() < 0.5 and n >= k:  # centroids drawn from the data (exact zeros, ties)
        cb = np.stack([X[rng.choice(n, k, replace=False), s * sd:(s + 1) * sd] for s in range(m)])
    else:
        cb = _draw_data(rng, m * k, sd, kind).reshape(m, k, sd)
    if rng.random() < 0.3 and k > 3:
        cb[:, k - 1] = cb[:, 0]  # duplicate centroid: the lower index must win
    cb = np.ascontiguousarray(cb, F)
    enc = _lib.PQEncoder(cb, metric)
    engines = [_lib.ENGINE_AUTO, _lib.ENGINE_EXACT]
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    for engine in engines:
        enc.set_engine(engine)
        codes, f16 = enc.encode(X)
        np.testing.assert_array_equal(codes.astype(np.uint32), want_c, err_msg=f"sd={sd} m={m} k={k} n={n} {kind} metric={metric} engine={engine}")
        same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
        assert same.all()
    enc.close()


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_fuzz_lloyd_step(oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    sd = int(rng.choice([2, 4, 6, 7, 8, 10, 12, 14, 16, 20, 24, 30, 31, 32, 40, 48, 50, 64, 70, 96, 128]))
    m = int(rng.choice([1, 2, 4, 8]))
    d = sd * m
    k = int(rng.choice([2, 16, 50, 128, 230, 256, 256]))
    n = int(rng.integers(k, 6000))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    X = _draw_data(rng, n, d, kind)
    init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.set_exact_update(True)  # reference-order sums: centroids bit-identical too
    km.init_from_rows(init)
    counts, changed = km.step()
    cent = km.get_centroids()
    assign = km.get_assignments()
    for s in range(m):
        c0 = X[init[s].astype(np.int64), s * sd:(s + 1) * sd]
        c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c0, threads=0)
        np.testing.assert_array_equal(assign[:, s].astype(np.uint32), a_ref, err_msg=f"sd={sd} m={m} k={k} n={n} {kind}")
        np.testing.assert_array_equal(counts[s], n_ref)
        ne = n_ref > 0
        assert cent[s][ne].tobytes() == c1[ne].tobytes()
        assert bool(changed[s]) == ch_ref
    km.close()
    ds.close()


WIDE_K = [257, 300, 511, 512, 777, 1024, 1500, 2048, 3000]


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_fuzz_pq_encode_wide(oracle, seed):
    """k > 256: two-byte codes; the grouped X32 screen (AUTO) and the exact engine against the oracle"""
    rng = np.random.default_rng(5000 + seed)
    sd = int(rng.choice([3, 4, 8, 12, 16, 16, 24, 32, 48, 64]))
    m = int(rng.choice([1, 2, 4]))
    d = sd * m
    k = int(rng.choice(WIDE_K))
    n = int(rng.integers(1, 1500))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    metric = int(rng.integers(0, 4))
    X = _draw_data(rng, n, d, kind)
    if rng.random() < 0.5 and n >= k:
        cb = np.stack([X[rng.choice(n, k, replace=False), s * sd:(s + 1) * sd] for s in range(m)])
    else:
        cb = _draw_data(rng, m * k, sd, kind).reshape(m, k, sd)
    if rng.random() < 0.3:
        cb[:, k - 1] = cb[:, 0]
    cb = np.ascontiguousarray(cb, F)
    enc = _lib.PQEncoder(cb, metric)
    want_c, want_f = oracle.pq_encode(metric, X, cb, threads=0)
    for engine in (_lib.ENGINE_AUTO, _lib.ENGINE_EXACT):
        enc.set_engine(engine)
        codes, f16 = enc.encode(X)
        assert codes.dtype == np.uint16
        np.testing.assert_array_equal(codes.astype(np.uint32), want_c, err_msg=f"sd={sd} m={m} k={k} n={n} {kind} metric={metric} engine={engine}")
        same = (f16.view(np.uint16) == want_f) | (np.isnan(f16) & np.isnan(want_f.view(np.float16)))
        assert same.all()
    enc.close()


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_fuzz_lloyd_step_wide(oracle, seed):
    rng = np.random.default_rng(6000 + seed)
    sd = int(rng.choice([2, 4, 8, 12, 16, 24, 32, 48, 64, 128]))
    m = int(rng.choice([1, 2, 4]))
    d = sd * m
    k = int(rng.choice(WIDE_K))
    n = int(rng.integers(k, 3 * k + 2000))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    exact = bool(rng.random() < 0.5)
    X = _draw_data(rng, n, d, kind)
    init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.set_exact_update(exact)
    km.init_from_rows(init)
    counts, changed = km.step()
    cent = km.get_centroids()
    assign = km.get_assignments()
    for s in range(m):
        c0 = X[init[s].astype(np.int64), s * sd:(s + 1) * sd]
        c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c0, threads=0)
        msg = f"sd={sd} m={m} k={k} n={n} {kind} exact={exact}"
        np.testing.assert_array_equal(assign[:, s].astype(np.uint32), a_ref, err_msg=msg)
        np.testing.assert_array_equal(counts[s], n_ref, err_msg=msg)
        ne = n_ref > 0
        if exact:
            assert cent[s][ne].tobytes() == c1[ne].tobytes(), msg
            assert bool(changed[s]) == ch_ref
        else:  # blocked f32 partial sums combined in f64: DESIGN.md "centroid tolerance"
            # (relative to the column's magnitude: the summands, not the mean, set the rounding error)
            with np.errstate(all="ignore"):
                scale = np.maximum(1.0, np.abs(X[:, s * sd:(s + 1) * sd]).max(axis=0))[None, :]
                err = np.abs(cent[s][ne] - c1[ne]) / scale
            fin = np.isfinite(c1[ne])
            assert (err[fin] <= 1e-5).all(), msg
    km.close()
    ds.close()


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_fuzz_tsvq(oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    d = int(rng.choice([1, 3, 8, 24, 32, 64, 128]))
    n = int(rng.integers(1, 5000))
    depth = int(rng.integers(0, 10))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    metric = int(rng.integers(0, 4))
    X = _draw_data(rng, n, d, kind)
    Q = np.concatenate([_draw_data(rng, 700, d, kind), X[:200]])
    ds = _lib.Dataset.from_host(X)
    cent, left, right = build_tree(ds, depth)
    ds.close()
    want = oracle.tsvq_build(X, depth)
    np.testing.assert_array_equal(left, want["left"], err_msg=f"d={d} n={n} depth={depth} {kind}")
    np.testing.assert_array_equal(right, want["right"])
    assert cent.tobytes() == want["centroids"].tobytes()
    names = ["squared_euclidean", "euclidean", "manhattan", "cosine"]
    t = TSVQ.from_tree(cent, left, right, Distance(names[metric]))
    want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, want, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf, err_msg=f"d={d} n={n} depth={depth} {kind} metric={metric}")
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_fuzz_device_driven_run_equals_step_loop(seed):
    """vqhip_kmeans_run (fused update, decisions on the device, pauses for reseeds) against the host-driven step loop
    on random shapes -- fused and non-fused sub_dims, k from 2 to 256 --, data kinds, duplicate initial rows (empty
    clusters -> pauses) and iteration budgets: identical iteration counts and pause counts, each path reproducible run
    to run, codebooks BIT-EQUAL (round 6: the fused update's row chunks come from all m subspaces whether retired ones are
    gated -- the run -- or dropped from the list -- the step loop --, so both group the same rows into the same partial
    sums; before, the step loop re-packed the survivors over the waves and the two fits could part by an iteration where
    the convergence test hangs on the last bit: seed 117 at VQ_FUZZ_SCALE=40)."""
    rng = np.random.default_rng(9000 + seed)
    sd = int(rng.choice([4, 8, 12, 16, 24, 32, 10, 7]))
    m = int(rng.integers(1, 9))
    k = int(rng.choice([2, 5, 16, 64, 100, 256]))
    n = int(rng.integers(max(2 * k, 300), 30_000))
    d = m * sd
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    X = _draw_data(rng, n, d, kind)
    if rng.random() < 0.5:
        X = (np.round(X * 4) / 4).astype(F)  # coarse grid: early convergence of some subspaces
    init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
    for _ in range(int(rng.integers(0, 4))):  # duplicate initial centroids -> empty clusters in the first iteration
        s_, a_, b_ = int(rng.integers(0, m)), int(rng.integers(0, k)), int(rng.integers(0, k))
        if a_ != b_:
            X[init[s_, a_]] = X[init[s_, b_]]
    ds = _lib.Dataset.from_host(X)
    reseed = [[int(x) for x in rng.integers(0, n, 4096)] for _ in range(m)]
    max_iters = int(rng.integers(1, 25))

    def fit(use_run):
        km = _lib.KMeans(ds, m, k)
        km.init_from_rows(init)
        active = np.ones(m, bool)
        iters = np.zeros(m, np.int64)
        its = [iter(r) for r in reseed]
        pauses, done = 0, 0
        while done < max_iters and active.any():
            if use_run:
                it, counts, changed, paused = km.run(max_iters - done)
                iters += it
                done += max(1, int(it.max()))
            else:
                counts, changed = km.step()
                iters[active] += 1
                done += 1
                paused = bool(((counts == 0) & active[:, None]).any())
            if paused:
                pauses += 1
                for s, j in np.argwhere((counts == 0) & active[:, None]):
                    km.patch_from_row(int(s), int(j), next(its[s]))
            active &= changed.astype(bool)
            km.set_active(active)
        cb = km.get_centroids()
        km.close()
        return cb, iters, pauses

    cb_run, it_run, p_run = fit(True)
    cb_step, it_step, p_step = fit(False)
    cb_run2, it_run2, _ = fit(True)
    ds.close()
    msg = f"seed={seed} n={n} m={m} k={k} sd={sd} {kind} max_iters={max_iters}"
    assert it_run.tolist() == it_step.tolist() and p_run == p_step, msg
    assert cb_run.tobytes() == cb_run2.tobytes() and it_run.tolist() == it_run2.tolist(), "run is not reproducible: " + msg
    assert cb_run.tobytes() == cb_step.tobytes(), "run and step loop differ: " + msg


@pytest.mark.parametrize("seed", range(32 * SCALE))
def test_fuzz_tsvq_descent_screens(oracle, seed):
    """The screened descents of all four metrics (one or two screened sums per level, proven margins, exact
    continuation) on random widths -- instantiated and zero-padded ones --, depths, data kinds and query families:
    same distribution, rescaled by a power of ten, negated (cosines of the other sign), the tree's own centroids and
    their neighbours (ties, zero distances)."""
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 100, 128, 192, 256, 300, 384, 512, 768]))
    n = int(rng.integers(2, 4000))
    depth = int(rng.integers(1, 10))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    X = _draw_data(rng, n, d, kind)
    tree = oracle.tsvq_build(X, depth)
    cent = tree["centroids"]
    scale = F(10.0 ** int(rng.integers(-12, 13)))
    Q = np.concatenate([_draw_data(rng, 600, d, kind), X[:150], _draw_data(rng, 200, d, kind) * scale,
                        -_draw_data(rng, 200, d, kind), cent[:200], cent[:200] * F(1.0000001), cent[:100] * F(-1),
                        np.zeros((1, d), F)])
    names = ["squared_euclidean", "euclidean", "manhattan", "cosine"]
    for metric in rng.permutation(4)[:2]:
        metric = int(metric)
        t = TSVQ.from_tree(cent, tree["left"], tree["right"], Distance(names[metric]))
        want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, tree, threads=0)
        msg = f"seed={seed} d={d} n={n} depth={depth} {kind} metric={metric}"
        np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf, err_msg=msg)
        np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16, err_msg=msg)


@pytest.mark.parametrize("rep", range(SCALE))
@pytest.mark.parametrize("kind", KINDS + ["nan_rows", "huge_then_small", "half_ulps"])
@pytest.mark.parametrize("shape", [(40_000, 32, 4), (70_001, 64, 3), (120_000, 128, 2), (50_000, 100, 3), (45_000, 36, 3),
                                   (40_000, 4, 4)])  # the last three: a short last block of columns (d % 32 != 0)
def test_tsvq_build_long_nodes_bit_identical(oracle, kind, shape, rep):
    """Nodes of >= 16384 rows take the tile-parallel exact emulation of the sequential column sums
    (k_fs_*): the tree must still be the oracle's bit for bit on data that stresses the binade /
    parity logic -- mixed signs, cancellation, exact half-ulp addends, NaN, scale jumps."""
    n, d, depth = shape
    import zlib

    rng = np.random.default_rng(zlib.crc32(f"{kind}-{n}-{rep}".encode()))
    if kind == "nan_rows":
        X = rng.standard_normal((n, d)).astype(F)
        X[rng.integers(0, n, 7), rng.integers(0, d, 7)] = np.nan
    elif kind == "huge_then_small":
        X = rng.random((n, d), dtype=F)
        X[: n // 3] *= F(1e12)
        X[n // 3: 2 * n // 3] *= F(1e-9)
    elif kind == "half_ulps":  # sums of dyadic values: exact ties in every addition once s is large
        X = (rng.integers(0, 8, (n, d)) * 0.5).astype(F) + F(0.25) * (rng.integers(0, 2, (n, d))).astype(F)
    else:
        X = _draw_data(rng, n, d, kind)
    ds = _lib.Dataset.from_host(X)
    try:
        cent, left, right = build_tree(ds, depth)
    except _lib.FfiError as e:  # the reference panics when a split column is all NaN
        assert "panics" in str(e)
        return
    finally:
        ds.close()
    want = oracle.tsvq_build(X, depth)
    np.testing.assert_array_equal(left, want["left"], err_msg=f"{kind} {shape} rep={rep}")
    np.testing.assert_array_equal(right, want["right"])
    a, b = cent, want["centroids"]
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{(~same).sum()} centroid components differ ({kind} {shape})"


_SEQ_WORKER = r'''
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.environ["VQ_REPO"]); sys.path.insert(0, os.path.join(os.environ["VQ_REPO"], "tests"))
from vq_amd import _lib
from vq_amd.tsvq import build_tree
from test_gpu_fuzz import _draw_data, KINDS
_lib.load(); _lib.set_device(0)
out = {}
for kind in KINDS:
    for n, d, depth in ((300_000, 64, 3), (1_000_000, 32, 2)):
        rng = np.random.default_rng(zlib.crc32(f"seq-{kind}-{n}".encode()))
        X = _draw_data(rng, n, d, kind)
        ds = _lib.Dataset.from_host(X)
        cent, left, right = build_tree(ds, depth)
        ds.close()
        out[f"{kind}-{n}-c"] = cent; out[f"{kind}-{n}-l"] = left; out[f"{kind}-{n}-r"] = right
np.savez(sys.argv[1], **out)
print("WORKER_OK")
'''


def test_tsvq_build_fast_sums_equal_sequential_kernel(tmp_path):
    """At sizes the CPU oracle is too slow for: the tile-parallel exact column sums (default) against
    the plain sequential kernel (VQHIP_TSVQ_SEQSUM=1), same library, 8 data kinds, up to 1M rows."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_SEQ_WORKER)
    res = {}
    for tag, env_extra in (("fast", {}), ("seq", {"VQHIP_TSVQ_SEQSUM": "1"})):
        out = tmp_path / f"{tag}.npz"
        env = dict(os.environ, VQ_REPO=root, **env_extra)
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0 and "WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        res[tag] = np.load(out)
    for key in res["fast"].files:
        a, b = res["fast"][key], res["seq"][key]
        assert a.shape == b.shape, key
        if a.dtype == np.float32:
            same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"{key}: {(~same).sum()} components differ"
        else:
            np.testing.assert_array_equal(a, b, err_msg=key)
