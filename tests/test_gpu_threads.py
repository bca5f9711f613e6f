"""One quantizer, many threads.

The reference's quantizers are plain data -- `ProductQuantizer` (src/pq.rs:39-45) and `TSVQ`
(src/tsvq.rs:186-191) are `Send + Sync`, `quantize(&self)` (src/pq.rs:167-199, src/tsvq.rs:239-255)
may run on any number of threads at once, and `Vector` is built for rayon (src/core/vector.rs:22-23).
pyvq serialises under the GIL (pyvq/src/pq.rs:96-107); this package calls through ctypes, which
RELEASES the GIL, so the guarantee has to come from libvqhip: one lock per handle, call-owned staging on
the per-vector path, stream order across threads (vq_amd/csrc/api.hip "handles under threads").

Every result below is compared bit for bit with the CPU oracle."""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle as O
from vq_amd import _lib

pytestmark = pytest.mark.gpu
F = np.float32
THREADS, CALLS = 8, 2000


def _codebooks(rng, m, k, sd):
    return rng.random((m, k, sd), dtype=F)


def _run_threads(fn, n_threads=THREADS):
    """fn(thread index) on n_threads threads released together; re-raises the first failure"""
    start = threading.Barrier(n_threads)

    def body(t):
        start.wait()
        return fn(t)

    with ThreadPoolExecutor(n_threads) as ex:
        return [f.result() for f in [ex.submit(body, t) for t in range(n_threads)]]


@pytest.mark.parametrize("metric", [_lib.EUCLIDEAN, _lib.COSINE])
def test_one_product_quantizer_eight_threads(oracle, metric):
    import vq_amd as pyvq

    rng = np.random.default_rng(11)
    m, k, sd = 4, 16, 8
    cb = _codebooks(rng, m, k, sd)
    dist = pyvq.Distance.euclidean() if metric == _lib.EUCLIDEAN else pyvq.Distance.cosine()
    pq = pyvq.ProductQuantizer.from_codebooks(cb, dist)
    X = rng.random((THREADS * CALLS, m * sd), dtype=F) - F(0.25)
    _, want = oracle.pq_encode(metric, X, cb)
    got = np.zeros_like(want)

    def work(t):
        for i in range(t * CALLS, (t + 1) * CALLS):
            got[i] = pq.quantize(X[i]).view(np.uint16)

    _run_threads(work)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, f"{bad.size} of {len(X)} vectors differ from the oracle, first rows {bad[:8]}"


def test_one_tsvq_eight_threads(oracle):
    import vq_amd as pyvq

    rng = np.random.default_rng(12)
    n, d = 6000, 32
    T = rng.random((n, d), dtype=F)
    tree = oracle.tsvq_build(T, 6)
    t = pyvq.TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], pyvq.Distance.squared_euclidean())
    X = rng.random((THREADS * CALLS, d), dtype=F)
    _, want = oracle.tsvq_encode(O.SQUARED_EUCLIDEAN, X, tree)
    got = np.zeros_like(want)

    def work(th):
        for i in range(th * CALLS, (th + 1) * CALLS):
            got[i] = t.quantize(X[i]).view(np.uint16)

    _run_threads(work)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, f"{bad.size} of {len(X)} vectors differ from the oracle, first rows {bad[:8]}"


def test_batch_and_vector_calls_mixed_on_one_encoder(oracle):
    """batch encodes (the handle's workspaces, MFMA screen) and per-vector calls (call-owned staging) from different
    threads on ONE encoder: the lock + the cross-stream order keep both exact"""
    rng = np.random.default_rng(13)
    m, k, sd = 8, 256, 16
    cb = _codebooks(rng, m, k, sd)
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    nb = 20000
    Xb = [rng.random((nb, m * sd), dtype=F) for _ in range(4)]
    Xv = rng.random((4 * 300, m * sd), dtype=F)
    want_b = [oracle.pq_encode(O.SQUARED_EUCLIDEAN, x, cb, threads=8) for x in Xb]
    want_v = oracle.pq_encode(O.SQUARED_EUCLIDEAN, Xv, cb, threads=8)

    def work(t):
        if t < 4:
            for _ in range(6):
                codes, f16 = enc.encode(Xb[t])
                assert np.array_equal(codes.astype(np.uint32), want_b[t][0])
                assert np.array_equal(f16.view(np.uint16), want_b[t][1])
        else:
            j = t - 4
            for i in range(j * 300, (j + 1) * 300):
                codes, f16 = enc.encode(Xv[i:i + 1])
                assert np.array_equal(codes[0].astype(np.uint32), want_v[0][i])
                assert np.array_equal(f16.view(np.uint16)[0], want_v[1][i])

    _run_threads(work)


def test_eight_encoders_one_shared_dataset(oracle):
    """distinct handles never share anything mutable: eight k-means fits + encoders over ONE resident data set, each on
    its own thread (the Rust shim's `&[&[f32]]` is shared by reference the same way)"""
    rng = np.random.default_rng(14)
    n, d, m, k = 12000, 32, 4, 16
    sd = d // m
    X = rng.random((n, d), dtype=F)
    ds = _lib.Dataset.from_host(X)
    inits = [np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64) for _ in range(THREADS)]
    want = []
    for t in range(THREADS):
        cbs = []
        for s in range(m):
            c0 = X[inits[t][s].astype(np.int64), s * sd:(s + 1) * sd]
            c1, a_ref, n_ref, _ = oracle.lloyd_step(X[:, s * sd:(s + 1) * sd], c0)
            cbs.append((c1, a_ref, n_ref))
        want.append(cbs)

    def work(t):
        km = _lib.KMeans(ds, m, k)
        km.init_from_rows(inits[t])
        counts, _ = km.step()
        assign = km.get_assignments()
        cb = km.get_centroids()
        for s in range(m):
            c1, a_ref, n_ref = want[t][s]
            assert np.array_equal(assign[:, s].astype(np.uint32), a_ref)
            assert np.array_equal(counts[s], n_ref)
            assert np.max(np.abs(cb[s] - c1) / np.maximum(1.0, np.abs(c1))) <= 1e-5
        km.close()
        ref_cb = np.stack([want[t][s][0] for s in range(m)])
        enc = _lib.PQEncoder(ref_cb, _lib.EUCLIDEAN)
        codes, f16 = enc.encode(X)
        rc, rf = oracle.pq_encode(O.EUCLIDEAN, X, ref_cb)
        assert np.array_equal(codes.astype(np.uint32), rc)
        assert np.array_equal(f16.view(np.uint16), rf)
        enc.close()

    _run_threads(work)
    ds.close()


def test_device_calls_from_two_threads_on_one_encoder(oracle):
    """asynchronous `_device` calls return with work queued on the CALLING thread's stream; the next call on the same
    handle from another thread (another stream) is ordered behind it by the library"""
    import torch

    rng = np.random.default_rng(15)
    m, k, sd = 8, 256, 16
    n = 200000
    cb = _codebooks(rng, m, k, sd)
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    Xs = [rng.random((n, m * sd), dtype=F) for _ in range(2)]
    want = [oracle.pq_encode(O.SQUARED_EUCLIDEAN, x, cb, want_f16=False, threads=8)[0] for x in Xs]
    dev = torch.device("cuda:0")
    xd = [torch.from_numpy(x).to(dev) for x in Xs]
    out = [torch.zeros((n, m), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    turn = [threading.Semaphore(1), threading.Semaphore(0)]

    def work(t):
        for _ in range(10):
            turn[t].acquire()  # strict alternation: every call finds the other thread's work still queued
            enc.encode_device(xd[t].data_ptr(), n, out[t].data_ptr(), None)
            turn[1 - t].release()
        _lib.synchronize()

    _run_threads(work, 2)
    torch.cuda.synchronize()
    for t in range(2):
        assert np.array_equal(out[t].cpu().numpy().astype(np.uint32), want[t])


def test_stats_after_the_handle_died_on_another_thread():
    """vqhip_last_assign_stats names the calling thread's last pass; the handle behind it may be destroyed by any
    thread meanwhile (a registry of live workspaces, not a dangling pointer)"""
    rng = np.random.default_rng(16)
    cb = _codebooks(rng, 8, 256, 16)
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    enc.encode(rng.random((5000, 128), dtype=F))
    rechecked, engine = _lib.last_assign_stats()
    assert engine == _lib.ENGINE_MFMA_BF16 and rechecked >= 0
    th = threading.Thread(target=enc.close)
    th.start()
    th.join()
    assert _lib.last_assign_stats() == (0, _lib.ENGINE_MFMA_BF16)


def test_large_host_batches_go_through_transfer_lanes(oracle):
    """host batches of 96 MB and more whose results are at least a quarter of the rows in bytes (f16 reconstructions) are
    cut into 32 MB chunks that three host threads of the LIBRARY carry end to end (H2D straight from the caller's rows,
    kernels, D2H straight into the caller's buffers), each on its own stream through the one encoder / tree handle:
    same bits as the oracle.  `xfer_lane_calls` proves which path a call took (PQ f16 and TSVQ f16: lanes; codes or
    leaf ids alone: the one-stream path)."""
    import vq_amd as pyvq

    rng = np.random.default_rng(17)
    m, k, sd = 8, 256, 16
    n = 230_000  # 118 MB of rows: several chunks per lane, a ragged last one
    cb = _codebooks(rng, m, k, sd)
    X = rng.random((n, m * sd), dtype=F)
    want_c, want_f = oracle.pq_encode(O.EUCLIDEAN, X, cb, threads=8)
    enc = _lib.PQEncoder(cb, _lib.EUCLIDEAN)
    lanes0 = _lib.xfer_lane_calls()
    codes, f16 = enc.encode(X)
    assert _lib.xfer_lane_calls() == lanes0 + 1
    assert np.array_equal(codes.astype(np.uint32), want_c) and np.array_equal(f16.view(np.uint16), want_f)
    rechecked, engine = _lib.last_assign_stats()
    assert engine == _lib.ENGINE_MFMA_BF16
    only_f16 = enc.encode(X, want_codes=False)[1]
    assert np.array_equal(only_f16.view(np.uint16), want_f)
    lanes1 = _lib.xfer_lane_calls()
    only_codes = enc.encode(X, want_f16=False)[0]
    assert _lib.xfer_lane_calls() == lanes1  # codes alone are 1.6 % of the rows: one stream
    assert np.array_equal(only_codes.astype(np.uint32), want_c)
    ds = _lib.Dataset.from_host(X)
    assert np.array_equal(ds.read(), X)
    tree = oracle.tsvq_build(X[:20000, :32].copy(), 6)
    t = pyvq.TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], pyvq.Distance.euclidean())
    Y = rng.random((900_000, 32), dtype=F)  # 115 MB: vqhip_tsvq_encode's run_lanes branch (three streams share the
    want_l, want_t = oracle.tsvq_encode(O.EUCLIDEAN, Y, tree, threads=8)  # undecided list and its turn counters)
    assert np.array_equal(t.leaf_ids(Y), want_l)
    lanes2 = _lib.xfer_lane_calls()
    assert np.array_equal(t.quantize_batch(Y).view(np.uint16), want_t)
    assert _lib.xfer_lane_calls() == lanes2 + 1
    assert np.array_equal(t.quantize_batch(Y).view(np.uint16), want_t)  # and again over the recycled lanes
    ds.close()
