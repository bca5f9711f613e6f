"""Small problems: one launch per Lloyd iteration (vq_amd/csrc/k_lloyd_small.hip).

The reference's own sizes (tests/integration_tests.rs:367-382, pyvq/tests/test_integrations.py:175-197) and BASELINE
configs[0] (10k x 64, m = 4, k = 16) take this path in vqhip_kmeans_step / vqhip_kmeans_run, the split forms
(accumulate / finalize, row-sharded) its slab variant.  Held here:
  * against the oracle: codes, counts, changed flags exact, centroids within the stated tolerance, per step and over a
    whole run with reseeds (same iteration counts);
  * the three forms against each other: step, run and accumulate + finalize give the SAME BITS;
  * against the general path (VQHIP_SMALL_LLOYD=0, a child process): codes / counts / flags / iteration counts equal,
    centroids within 1e-6 relative (other row ranges behind the f64 combination).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from vq_amd import _lib

pytestmark = pytest.mark.gpu
F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [  # (n, d, m, k)
    (10_000, 64, 4, 16),   # BASELINE configs[0]
    (3000, 32, 4, 32),
    (257, 10, 2, 4),       # sub_dim 5, one row past a 256-row range
    (1000, 21, 3, 7),      # sub_dim 7
    (32768, 8, 1, 16),     # the longest the path takes, m = 1 (lbg_quantize on whole vectors): 512 ranges x 128 partials
    (5000, 96, 3, 32),     # k * sub_dim = 1024
    (100, 4, 4, 3),        # sub_dim 1
]


def _init(n, m, k, seed=3):
    rng = np.random.default_rng(seed)
    return np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)


@pytest.mark.parametrize("shape", SHAPES)
def test_step_against_the_oracle(oracle, shape):
    n, d, m, k = shape
    sd = d // m
    X = np.random.default_rng(1).random((n, d), dtype=F)
    init = _init(n, m, k)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(init)
    cb = np.stack([X[init[s].astype(np.int64), s * sd:(s + 1) * sd] for s in range(m)])
    for _ in range(3):
        counts, changed = km.step()
        assign = km.get_assignments()
        got = km.get_centroids()
        assert _lib.last_assign_stats()[1] == _lib.ENGINE_EXACT  # the one-launch path reports the exact engine
        for s in range(m):
            c1, a_ref, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], cb[s])
            np.testing.assert_array_equal(assign[:, s].astype(np.uint32), a_ref)
            np.testing.assert_array_equal(counts[s], n_ref)
            assert bool(changed[s]) == ch_ref
            assert np.max(np.abs(got[s] - c1) / np.maximum(1.0, np.abs(c1))) <= 1e-5
        cb = got  # the next step starts from the library's centroids on both sides
    km.close()
    ds.close()


def test_run_step_and_split_forms_give_the_same_bits():
    n, d, m, k = 10_000, 64, 4, 16
    X = np.random.default_rng(2).random((n, d), dtype=F)
    init = _init(n, m, k, seed=4)
    ds = _lib.Dataset.from_host(X)

    def fit(form):
        km = _lib.KMeans(ds, m, k)
        km.init_from_rows(init)
        iters = np.zeros(m, np.int64)
        if form == "run":
            it, counts, changed, paused = km.run(12)
            assert not paused
            iters = it.astype(np.int64)
        else:
            active = np.ones(m, bool)
            for _ in range(12):
                if not active.any():
                    break
                if form == "step":
                    counts, changed = km.step()
                else:
                    km.accumulate()
                    counts, changed = km.finalize()
                iters += active
                assert (counts[active] > 0).all()
                active &= changed
                km.set_active(active)
        out = (km.get_centroids(), km.get_assignments(), iters)
        km.close()
        return out

    run, step, split = fit("run"), fit("step"), fit("split")
    for a, b in ((run, step), (run, split)):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[2], b[2])
    # (assignments of a retired subspace stay those of its last iteration in every form)
    np.testing.assert_array_equal(run[1], step[1])
    ds.close()


def test_whole_fit_with_reseeds_matches_the_oracle(oracle):
    """0/1 lattice rows (16 distinct sub-vectors per subspace, k = 16 centroids drawn from rows with repeats): empty
    clusters -> the run pauses, the host patches in the injected reseed rows, subspaces retire at different iterations
    -- the control flow of src/core/vector.rs:415-458 end to end.  Every sum of such rows is exact in f32 whatever its
    order, so the `changed` flags cannot depend on the summation structure and the trajectory is the oracle's bit for bit."""
    import vq_amd as pyvq

    n, d, m, k, iters = 4000, 16, 4, 16, 10
    rng = np.random.default_rng(5)
    X = rng.integers(0, 2, (n, d)).astype(F)
    init = _init(n, m, k, seed=6)
    reseed = rng.integers(0, n, (m, 400)).astype(np.uint64)
    want_cb, want_iters = oracle.pq_fit(X, m, k, iters, init, reseed, threads=0)
    pq = pyvq.ProductQuantizer(X, m, k, iters, pyvq.Distance.euclidean(), 42, init_rows=init, reseed_rows=reseed)
    assert pq.fit_stats["reseeds"] > 0  # the path under test
    np.testing.assert_array_equal(pq.fit_stats["iters"], want_iters)
    np.testing.assert_array_equal(pq.codebooks, want_cb)
    codes = pq.encode(X)
    want_c, _ = oracle.pq_encode(O.EUCLIDEAN, X, pq.codebooks, want_f16=False, threads=0)
    np.testing.assert_array_equal(codes.astype(np.uint32), want_c)


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VQ_REPO"])
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
out = {}
for name, (n, d, m, k) in {"c1": (10000, 64, 4, 16), "odd": (1000, 21, 3, 7)}.items():
    X = np.random.default_rng(7).random((n, d), dtype=np.float32)
    rng = np.random.default_rng(8)
    init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(init)
    for it in range(3):
        counts, changed = km.step()
        out[f"{name}_counts{it}"] = counts
        out[f"{name}_changed{it}"] = changed
        out[f"{name}_assign{it}"] = km.get_assignments()
        out[f"{name}_cb{it}"] = km.get_centroids()
    km.init_from_rows(init)
    km.set_active(np.ones(m, np.uint8))
    itr, counts, changed, paused = km.run(8)
    out[name + "_run_iters"] = itr
    out[name + "_run_cb"] = km.get_centroids()
    out[name + "_engine"] = np.array(_lib.last_assign_stats()[1])
    km.close(); ds.close()
# an early-converging run (0/1 lattice rows: every sum is exact in f32, both paths walk the same trajectory): iterations
# stay queued behind the retirement of the last subspace, and the counts / changed flags handed back are those of the
# last EXECUTED iteration in both forms (include/vqhip.h; ADVICE r4: the small path zeroed them)
n, d, m, k = 4000, 16, 4, 4
X = np.random.default_rng(5).integers(0, 2, (n, d)).astype(np.float32)
rng = np.random.default_rng(6)
init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
ds = _lib.Dataset.from_host(X)
km = _lib.KMeans(ds, m, k)
km.init_from_rows(init)
itr, counts, changed, paused = km.run(40)
out["lattice_run_iters"] = itr
out["lattice_run_counts"] = counts
out["lattice_run_changed"] = changed
out["lattice_run_paused"] = np.array(paused)
out["lattice_run_active"] = km.get_active()
out["lattice_run_cb"] = km.get_centroids()
km.close(); ds.close()
np.savez(sys.argv[1], **out)
print("WORKER_OK")
'''


def test_against_the_general_path(tmp_path):
    def run(flag):
        script = tmp_path / "w.py"
        script.write_text(_WORKER)
        out = tmp_path / f"s{flag}.npz"
        env = dict(os.environ, VQHIP_SMALL_LLOYD=str(flag), VQ_REPO=ROOT)
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        return np.load(out)

    small, general = run(1), run(0)
    if not bool(small["lattice_run_paused"]):  # converged before the 40 queued iterations ran out: the case under test
        assert int(small["lattice_run_iters"].max()) < 40
        last = small["lattice_run_iters"] == small["lattice_run_iters"].max()
        assert (small["lattice_run_counts"][last].sum(axis=1) == 4000).all()  # the last executed iteration's counts survive
    assert int(small["c1_engine"]) == _lib.ENGINE_EXACT and int(general["c1_engine"]) == _lib.ENGINE_MFMA_BF16
    for key in small.files:
        if key.endswith("_engine"):
            continue
        if "_cb" in key:
            a, b = small[key], general[key]
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= 1e-6, key
        else:
            np.testing.assert_array_equal(small[key], general[key], err_msg=key)
