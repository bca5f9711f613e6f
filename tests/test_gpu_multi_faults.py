"""Failure containment and the remaining row-block paths of the one-process multi-GPU handles ("next round" 3):
the exchange self-test behind vqhip_comm_create_local, bounded waits, a rank that fails in the middle of a run (fault
injection compiled into the library, off unless the VQHIP_TEST_* variables are set), the constructor's device defaults,
and TSVQ encode / decode / dequantize in row blocks over device slots (src/tsvq.rs:239-265, src/pq.rs:201-209).

One-GPU boxes: the device list names device 0 several times (every slot is a rank of its own)."""
import os
import time

import numpy as np
import pytest

import oracle as O
from vq_amd import _lib
from vq_amd.errors import FfiError

pytestmark = pytest.mark.gpu
F = np.float32


def _init(n, m, k):
    return np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)


class _env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _healthy_fit(X, m, k, slots):
    mds = _lib.MDataset.from_host(X, [0] * slots)
    km = _lib.MKMeans(mds, m, k)
    km.init_from_rows(_init(X.shape[0], m, k))
    it, counts, _, _ = km.run(3)
    km.close()
    mds.close()
    assert int(counts.sum()) == m * X.shape[0]


def test_exchange_selftest_names_the_device_pair():
    n, d, m, k = 50_000, 64, 4, 32
    X = np.random.default_rng(1).random((n, d), dtype=F)
    mds = _lib.MDataset.from_host(X, [0, 0, 0])
    with _env(VQHIP_TEST_SELFTEST_CORRUPT=1):
        t0 = time.time()
        with pytest.raises(FfiError) as ei:
            _lib.MKMeans(mds, m, k)
        assert time.time() - t0 < 20
    msg = str(ei.value)
    assert "exchange self-test" in msg and "rank 1" in msg and "device 0" in msg, msg
    # the group died with the constructor; the data set is intact and a new k-means handle passes its self-test
    km = _lib.MKMeans(mds, m, k)
    assert km.info() == (3, 2)
    km.init_from_rows(_init(n, m, k))
    _, counts, _, _ = km.run(2)
    assert int(counts.sum()) == m * n
    km.close()
    mds.close()


@pytest.mark.parametrize("shape", [(300_000, 64, 4, 256), (60_000, 64, 4, 24)])  # device-driven run loop / host-driven loop
def test_a_failing_rank_releases_its_peers(shape):
    n, d, m, k = shape
    X = np.random.default_rng(2).random((n, d), dtype=F)
    mds = _lib.MDataset.from_host(X, [0, 0, 0])
    km = _lib.MKMeans(mds, m, k)
    km.init_from_rows(_init(n, m, k))
    with _env(VQHIP_TEST_FAIL_RANK=1, VQHIP_TEST_FAIL_ITER=2):
        t0 = time.time()
        with pytest.raises(FfiError) as ei:
            km.run(10)
        took = time.time() - t0
    assert took < 20, took  # (the default timeout is 60 s: the peers were woken, they did not time out)
    msg = str(ei.value)
    assert "fault injection" in msg and "rank 1" in msg, msg
    assert ei.value.status == _lib.ERR_RUNTIME
    # the group stays poisoned: later collectives fail at once, with the first failure's text
    t0 = time.time()
    with pytest.raises(FfiError) as ei2:
        km.run(1)
    assert time.time() - t0 < 5 and "poisoned" in str(ei2.value) and "fault injection" in str(ei2.value), str(ei2.value)
    km.close()  # handles destroy cleanly
    mds.close()
    _healthy_fit(X[:50_000], m, min(k, 32), 2)  # and the process goes on


def test_a_rank_that_dies_silently_times_its_peers_out():
    n, d, m, k = 100_000, 64, 4, 32
    X = np.random.default_rng(3).random((n, d), dtype=F)
    with _env(VQHIP_COMM_TIMEOUT_S=2, VQHIP_TEST_FAIL_SILENT=1, VQHIP_TEST_FAIL_RANK=0, VQHIP_TEST_FAIL_ITER=1):
        mds = _lib.MDataset.from_host(X, [0, 0])
        km = _lib.MKMeans(mds, m, k)  # (the group reads its timeout when it is made)
        km.init_from_rows(_init(n, m, k))
        t0 = time.time()
        with pytest.raises(FfiError) as ei:
            km.run(5)
        took = time.time() - t0
    assert 1.5 <= took < 30, took
    assert "fault injection" in str(ei.value) or "waited" in str(ei.value), str(ei.value)
    with pytest.raises(FfiError) as ei2:
        km.run(1)
    assert "waited 2 s" in str(ei2.value) and "VQHIP_COMM_TIMEOUT_S" in str(ei2.value), str(ei2.value)
    km.close()
    mds.close()


def test_comm_abort_wakes_a_blocked_rank():
    """vqhip_comm_abort from another thread: a rank alone in a two-rank collective returns at once"""
    import ctypes as C
    import threading

    lib = _lib.load()
    grp = C.c_void_p()
    _lib.check(lib.vqhip_comm_group_create(2, C.byref(grp)))
    comms = [C.c_void_p(), C.c_void_p()]
    rcs = [None, None]

    def make(r):
        _lib.set_device(0)
        rcs[r] = lib.vqhip_comm_create_local(grp, r, C.byref(comms[r]))

    th = [threading.Thread(target=make, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert rcs == [0, 0]
    X = np.random.default_rng(4).random((20_000, 32), dtype=F)
    ds = _lib.Dataset.from_host(X)
    km = _lib.KMeans(ds, 2, 16)
    km.init_from_rows(_init(20_000, 2, 16))
    out = {}

    def lone():  # rank 0 enters a sharded step; rank 1 never comes
        _lib.set_device(0)
        t0 = time.time()
        counts = np.empty((2, 16), np.uint32)
        changed = np.empty(2, np.uint8)
        out["rc"] = lib.vqhip_kmeans_step_sharded(km.raw, comms[0], _lib.ptr(counts, _lib._u32p), _lib.ptr(changed, _lib._u8p))
        out["err"] = _lib.last_error()
        out["took"] = time.time() - t0

    t = threading.Thread(target=lone)
    t.start()
    time.sleep(1.0)
    assert t.is_alive()  # blocked in the exchange's barrier
    _lib.check(lib.vqhip_comm_abort(comms[1]))
    t.join(10)
    assert not t.is_alive() and out["rc"] == _lib.ERR_RUNTIME and "poisoned" in out["err"], out
    km.close()
    ds.close()
    for c in comms:
        lib.vqhip_comm_destroy(c)
    lib.vqhip_comm_group_destroy(grp)


def test_a_rank_that_never_joins_times_the_constructor_out():
    """vqhip_comm_create_local with one of two ranks missing: the bounded wait ends the call, the group is poisoned, its
    slot is released by vqhip_comm_group_destroy"""
    import ctypes as C

    lib = _lib.load()
    with _env(VQHIP_COMM_TIMEOUT_S=1):
        grp = C.c_void_p()
        _lib.check(lib.vqhip_comm_group_create(2, C.byref(grp)))
    comm = C.c_void_p()
    _lib.set_device(0)
    t0 = time.time()
    rc = lib.vqhip_comm_create_local(grp, 0, C.byref(comm))
    took = time.time() - t0
    assert rc == _lib.ERR_RUNTIME and not comm.value and 0.8 <= took < 10, (rc, took)
    assert "waited 1 s" in _lib.last_error() and "1 of 2 ranks" in _lib.last_error(), _lib.last_error()
    rc2 = lib.vqhip_comm_create_local(grp, 1, C.byref(comm))  # the late rank finds the group poisoned: no wait
    assert rc2 == _lib.ERR_RUNTIME and "poisoned" in _lib.last_error()
    lib.vqhip_comm_group_destroy(grp)


def test_constructor_device_defaults(monkeypatch):
    """devices=None: every visible device the batch gives work to -- but ONE under exact_update (it raised)"""
    import vq_amd as pyvq

    n, d, m, k = 140_000, 64, 4, 16  # 8.96M elements: two devices' worth
    X = np.random.default_rng(5).random((n, d), dtype=F)
    init = _init(n, m, k)
    want = pyvq.ProductQuantizer(X, m, k, 3, None, 42, init_rows=init, exact_update=True, devices=[0])
    monkeypatch.setattr(_lib, "device_count", lambda: 4)
    got = pyvq.ProductQuantizer(X, m, k, 3, None, 42, init_rows=init, exact_update=True)  # devices=None
    assert got.fit_stats["devices"] == [0]
    np.testing.assert_array_equal(got.codebooks, want.codebooks)
    with pytest.raises(FfiError):  # an explicit list keeps the library's refusal
        pyvq.ProductQuantizer(X, m, k, 3, None, 42, init_rows=init, exact_update=True, devices=[0, 0])
    with pytest.raises(FfiError) as ei:  # and without exact_update the default IS several devices (device 1 does not exist here)
        pyvq.ProductQuantizer(X, m, k, 3, None, 42, init_rows=init)
    assert "device 1 out of range" in str(ei.value)


def test_one_slot_on_another_device_goes_through_its_worker(monkeypatch, oracle):
    """devices=[d] with d != the thread's current device: the one-slot multi-device handles (their worker thread owns the
    device), the single-device result bit for bit (the id was ignored)"""
    import vq_amd as pyvq

    n, d, m, k = 70_000, 64, 4, 16
    X = np.random.default_rng(6).random((n, d), dtype=F)
    init = _init(n, m, k)
    plain = pyvq.ProductQuantizer(X, m, k, 4, None, 42, init_rows=init, devices=[0])
    assert not plain._pinned and plain._menc is None
    monkeypatch.setattr(_lib, "get_device", lambda: 1)  # "the current device is another one"
    pinned = pyvq.ProductQuantizer(X, m, k, 4, None, 42, init_rows=init, devices=[0])
    assert pinned._pinned and pinned._menc is not None and pinned.fit_stats["devices"] == [0]
    np.testing.assert_array_equal(pinned.codebooks, plain.codebooks)
    assert np.array_equal(pinned.quantize(X[3]).view(np.uint16), plain.quantize(X[3]).view(np.uint16))
    assert np.array_equal(pinned.encode(X[:1000]), plain.encode(X[:1000]))


def test_bad_out_array_to_a_multi_device_quantizer():
    import vq_amd as pyvq

    n, d, m, k = 70_000, 32, 4, 16
    X = np.random.default_rng(7).random((n, d), dtype=F)
    pq = pyvq.ProductQuantizer(X, m, k, 2, None, 42, devices=[0, 0])
    good = np.empty((n, d), np.float16)
    assert pq.quantize_batch(X, out=good) is not None
    for bad in (np.empty((n - 1, d), np.float16), np.empty((n, d), np.float32), np.empty((n, 2 * d), np.float16)[:, ::2],
                np.empty((n, d), np.uint8)):
        with pytest.raises(FfiError):
            pq.quantize_batch(X, out=bad)
    ro = np.empty((n, d), np.float16)
    ro.flags.writeable = False
    with pytest.raises(FfiError):
        pq.quantize_batch(X, out=ro)


def test_decode_and_dequantize_row_blocks(oracle):
    n, m, k, sd = 200_001, 8, 256, 16
    rng = np.random.default_rng(8)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8)
    menc = _lib.MPQEncoder(cb, _lib.EUCLIDEAN, [0, 0, 0])
    got = menc.decode(codes)
    want = np.concatenate([cb[s][codes[:, s]] for s in range(m)], axis=1)
    np.testing.assert_array_equal(got, want)
    h = rng.standard_normal((n, m * sd)).astype(np.float16)
    h[0, :4] = [np.inf, -np.inf, np.nan, np.float16(6e-8)]
    out = menc.dequantize_f16(h)
    np.testing.assert_array_equal(out.view(np.uint32), h.astype(F).view(np.uint32))
    menc.close()
    import vq_amd as pyvq

    pq = pyvq.ProductQuantizer.from_codebooks(cb)
    np.testing.assert_array_equal(pq.dequantize_batch(h[:100]).view(np.uint32), h[:100].astype(F).view(np.uint32))


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_tsvq_row_blocks_over_slots(oracle, metric):
    import vq_amd as pyvq

    n, d, depth = 300_001, 64, 6
    rng = np.random.default_rng(9)
    X = rng.random((n, d), dtype=F)
    dist = getattr(pyvq.Distance, metric)()
    one = pyvq.TSVQ(X[:50_000], depth, dist, devices=[0])
    cen, lf, rt = one.tree
    multi = pyvq.TSVQ.from_tree(cen, lf, rt, dist, devices=[0, 0, 0])
    assert multi.devices == [0, 0, 0] and multi._menc is None  # (made by the first batch that uses it)
    leaf1, leaf3 = one.leaf_ids(X), multi.leaf_ids(X)
    assert multi._last_multi
    scr, und = multi.last_encode_stats()
    assert isinstance(scr, bool) and und >= 0
    np.testing.assert_array_equal(leaf1, leaf3)
    want_leaf, want_f16 = oracle.tsvq_encode(dist.metric, X, dict(centroids=cen, left=lf, right=rt), threads=0)
    np.testing.assert_array_equal(leaf3, want_leaf)
    q = multi.quantize_batch(X)
    np.testing.assert_array_equal(q.view(np.uint16), want_f16)
    np.testing.assert_array_equal(multi.dequantize_batch(q).view(np.uint32), q.astype(F).view(np.uint32))
    small = multi.leaf_ids(X[:1000])  # (below the hand-over threshold: the single-device encoder)
    assert not multi._last_multi
    np.testing.assert_array_equal(small, want_leaf[:1000])
