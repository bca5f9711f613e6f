#!/usr/bin/env python3
"""Times the hot entry points over unusual shapes (sub_dims without a fixed-length kernel, long vectors, odd tree
dimensions, all four metrics) and prints achieved algorithmic rates, to spot performance cliffs next to the
BASELINE shapes.  Device-resident inputs; one MI355X."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vq_amd import TSVQ, Distance, _lib  # noqa: E402
from vq_amd.tsvq import build_tree  # noqa: E402

_lib.load()
_lib.set_device(0)
lib = _lib.load()


def timed(fn, reps=3):
    fn()
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    _lib.synchronize()
    return (time.perf_counter() - t0) / reps


def pq_case(n, d, m, k, metric):
    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    rng = np.random.default_rng(1)
    cb = rng.random((m, k, d // m), dtype=np.float32)
    enc = _lib.PQEncoder(cb, metric)
    codes = torch.empty((n, m * (1 if k <= 256 else 2)), dtype=torch.uint8, device="cuda")
    dt = timed(lambda: enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None))
    _, engine = _lib.last_assign_stats()
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
    dk = timed(lambda: km.step(), reps=2)
    km.close()
    enc.close()
    ds.close()
    print(f"PQ   n={n:8d} d={d:5d} m={m:3d} k={k:5d} sd={d//m:5d} metric={metric} engine={engine} "
          f"encode {dt*1e3:9.3f} ms ({2.0*k*d*n/dt/1e12:6.1f} alg TFLOP/s)  kmeans step {dk*1e3:9.3f} ms", flush=True)


def tsvq_case(n, d, depth, name):
    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    tb = timed(lambda: build_tree(ds, depth), reps=2)
    cent, left, right = build_tree(ds, depth)
    t = TSVQ.from_tree(cent, left, right, Distance(name))
    leaf = torch.empty(n, dtype=torch.int32, device="cuda")
    te = timed(lambda: _lib.check(lib.vqhip_tsvq_encode_device(t._enc.raw, C.c_void_p(ds.device_ptr), n,
                                                               C.c_void_p(leaf.data_ptr()), None)))
    ds.close()
    print(f"TSVQ n={n:8d} d={d:5d} depth={depth} {name:18s} build {tb*1e3:9.3f} ms ({n*d*4*depth/tb/1e9:7.1f} GB/s/level-pass)  "
          f"encode {te*1e3:9.3f} ms ({n*d*4/te/1e9:7.1f} GB/s)", flush=True)


if __name__ == "__main__":
    for metric in (0, 1, 2, 3):
        pq_case(1_000_000, 128, 8, 256, metric)
    for (n, d, m, k) in ((1_000_000, 100, 10, 256), (1_000_000, 105, 15, 256), (1_000_000, 96, 32, 256),
                         (1_000_000, 128, 128, 16), (500_000, 300, 30, 100), (200_000, 384, 1, 256),
                         (100_000, 1536, 1, 64), (1_000_000, 128, 8, 1), (1_000_000, 128, 8, 3), (100, 128, 8, 16),
                         (1_000_000, 2, 1, 256), (1_000_000, 128, 1, 256)):
        for metric in (0, 3):
            pq_case(n, d, m, k, metric)
    for (n, d, depth) in ((1_000_000, 128, 8), (1_000_000, 100, 8), (500_000, 384, 6), (1_000_000, 384, 5), (200_000, 768, 6), (200_000, 768, 5), (100_000, 1024, 5), (1_000_000, 7, 10),
                          (1_000_000, 128, 12), (100_000, 128, 8)):
        for name in ("squared_euclidean", "cosine"):
            tsvq_case(n, d, depth, name)
