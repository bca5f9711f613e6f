"""One call, one process, several GPUs (include/vqhip.h "one call, one process, several GPUs"): the ranks
of the row-sharded fit as worker threads inside the library, behind the drop-in constructor (src/pq.rs:83-141 is ONE call).

The boxes of this pool have one GPU, so the device list names device 0 several times: every slot is a rank of its own
(thread, stream, row block, vqhip_kmeans, communicator) and the per-iteration exchange is the in-process fixed-order one.
What cannot run here is peer access between DIFFERENT devices and the RCCL variant with more than one rank."""
import numpy as np
import pytest

import oracle as O
from vq_amd import _lib
from vq_amd.errors import FfiError
from vq_amd.pq import fit_codebooks

pytestmark = pytest.mark.gpu
F = np.float32


def _init(n, m, k):
    return np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)


def test_one_slot_is_the_single_device_fit_bit_for_bit():
    n, d, m, k = 20000, 64, 4, 32
    X = np.random.default_rng(6).random((n, d), dtype=F)
    init = _init(n, m, k)
    ds = _lib.Dataset.from_host(X)
    want = fit_codebooks(ds, m, k, 6, init_rows=init)
    ds.close()
    mds = _lib.MDataset.from_host(X, [0])
    stats = {}
    got = fit_codebooks(mds, m, k, 6, init_rows=init, stats=stats)
    km = _lib.MKMeans(mds, m, k)
    assert km.info() == (1, 0)  # one rank: the identity communicator
    km.close()
    # the reference's summation order through the one-slot handles: the single-device bits
    exact = fit_codebooks(mds, m, k, 3, init_rows=init, exact_update=True)
    mds.close()
    ds = _lib.Dataset.from_host(X)
    np.testing.assert_array_equal(exact, fit_codebooks(ds, m, k, 3, init_rows=init, exact_update=True))
    ds.close()
    two = _lib.MDataset.from_host(X, [0, 0])
    with pytest.raises(FfiError):
        fit_codebooks(two, m, k, 3, init_rows=init, exact_update=True)
    two.close()
    np.testing.assert_array_equal(got, want)
    enc, menc = _lib.PQEncoder(want, _lib.EUCLIDEAN), _lib.MPQEncoder(want, _lib.EUCLIDEAN, [0])
    c0, f0 = enc.encode(X)
    c1, f1 = menc.encode(X)
    assert np.array_equal(c0, c1) and np.array_equal(f0.view(np.uint16), f1.view(np.uint16))
    enc.close()
    menc.close()


@pytest.mark.parametrize("slots", [2, 3, 4])
def test_several_slots_one_gpu_fit(oracle, slots):
    """uneven row blocks, duplicates among the initial rows (empty clusters: the run pauses on every rank alike and the
    reseed row is named by its GLOBAL id), ten iterations: iteration counts, reseeds and every cluster's count as in the
    single-device fit, centroids within 1e-6 (the f64 slab is summed in another grouping), bit-identical run to run"""
    n, d, m, k = 200_003, 64, 4, 32
    X = np.random.default_rng(6).random((n, d), dtype=F)
    X[17] = X[4]
    X[4000] = X[4]
    init = _init(n, m, k)
    init[0, 1], init[0, 2], init[0, 0] = 17, 4000, 4
    reseed = [[7, 9, 150_000, 13] * 8] * m
    ds = _lib.Dataset.from_host(X)
    s0 = {}
    want = fit_codebooks(ds, m, k, 10, init_rows=init, reseed_rows=reseed, stats=s0)
    ds.close()
    assert s0["reseeds"] > 0  # the path under test
    outs = []
    for _ in range(2):
        mds = _lib.MDataset.from_host(X, [0] * slots)
        assert mds.rows_per_device().tolist() == [_lib.shard_rows(n, slots, r)[1] for r in range(slots)]
        km = _lib.MKMeans(mds, m, k)
        assert km.info() == (slots, 2)  # the in-process exchange
        km.close()
        s1 = {}
        outs.append(fit_codebooks(mds, m, k, 10, init_rows=init, reseed_rows=reseed, stats=s1))
        mds.close()
        assert s1["reseeds"] == s0["reseeds"] and s1["iters"].tolist() == s0["iters"].tolist()
    np.testing.assert_array_equal(outs[0], outs[1])
    # ten iterations on, the two fits have parted at boundary rows (the f64 slab is summed in another grouping: last-bit
    # differences in a mean move a row across a boundary, and Lloyd amplifies that): same quality, not the same bits
    sd = d // m

    def inertia(cb):
        tot = 0.0
        for s in range(m):
            xs = X[:, s * sd:(s + 1) * sd]
            _, a, _, _ = oracle.lloyd_step(xs, cb[s], threads=0)
            diff = xs - cb[s][a]
            tot += float(np.einsum("ij,ij->", diff, diff, dtype=np.float64))
        return tot

    i_multi, i_one = inertia(outs[0]), inertia(want)
    assert abs(i_multi - i_one) <= 1e-3 * i_one, (i_multi, i_one)
    # one sharded Lloyd step against the oracle: global counts exact, centroids within the step tolerance
    mds = _lib.MDataset.from_host(X, [0] * slots)
    km = _lib.MKMeans(mds, m, k)
    km.set_centroids(want)
    it, counts, changed, paused = km.run(1)
    got = km.get_centroids()
    km.close()
    mds.close()
    for s in range(m):
        c1, _, n_ref, ch_ref = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], want[s], threads=0)
        np.testing.assert_array_equal(counts[s], n_ref)
        assert bool(changed[s]) == ch_ref
        assert np.max(np.abs(got[s] - c1) / np.maximum(1.0, np.abs(c1))) <= 1e-5


def test_encode_row_blocks_over_slots(oracle):
    n, m, k, sd = 300_001, 8, 256, 16
    rng = np.random.default_rng(8)
    X = rng.random((n, m * sd), dtype=F)
    cb = X[rng.choice(n, m * k, replace=False)].reshape(m, k, m * sd)[:, :, :sd].copy()
    want_c, want_f = oracle.pq_encode(O.EUCLIDEAN, X, cb, threads=0)
    menc = _lib.MPQEncoder(cb, _lib.EUCLIDEAN, [0, 0, 0])
    codes, f16 = menc.encode(X)
    assert np.array_equal(codes.astype(np.uint32), want_c) and np.array_equal(f16.view(np.uint16), want_f)
    # the resident rows of a sharded data set (what bench.py --one-process times)
    mds = _lib.MDataset.from_host(X, [0, 0, 0])
    resident = menc.encode_dataset(mds, repeat=2, want_codes=True)
    assert np.array_equal(resident.astype(np.uint32), want_c)
    mds.close()
    with pytest.raises(FfiError):
        other = _lib.MDataset.from_host(X[:1000], [0, 0])
        try:
            menc.encode_dataset(other)
        finally:
            other.close()
    menc.close()


def test_constructor_takes_a_device_list(oracle):
    """ProductQuantizer(X, m, k, ..., devices=[0, 0]): the reference's one call (src/pq.rs:83-141), two ranks inside"""
    import vq_amd as pyvq

    n, d, m, k = 150_000, 64, 4, 16
    X = np.random.default_rng(9).random((n, d), dtype=F)
    init = _init(n, m, k)
    one = pyvq.ProductQuantizer(X, m, k, 5, pyvq.Distance.euclidean(), 42, init_rows=init, devices=[0])
    two = pyvq.ProductQuantizer(X, m, k, 5, pyvq.Distance.euclidean(), 42, init_rows=init, devices=[0, 0])
    assert one.fit_stats["devices"] == [0] and two.fit_stats["devices"] == [0, 0]
    assert two.fit_stats["iters"].tolist() == one.fit_stats["iters"].tolist()
    assert np.max(np.abs(two.codebooks - one.codebooks)) <= 5e-2  # (same fit up to boundary rows: see above)
    q = two.quantize_batch(X)  # row blocks over both slots
    want_c, want_f = oracle.pq_encode(O.EUCLIDEAN, X, two.codebooks, threads=0)
    assert np.array_equal(q.view(np.uint16), want_f)
    assert np.array_equal(two.encode(X).astype(np.uint32), want_c)
    assert np.array_equal(two.quantize(X[5]).view(np.uint16), want_f[5])
    with pytest.raises(FfiError):
        pyvq.ProductQuantizer(X[:1000], m, k, 2, devices=[0, 7])
