"""C1 (10k x 64, m=4, k=16): a few Lloyd iterations and encode passes, for a kernel trace of the launch-bound regime."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = 10_000, 64, 4, 16
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
km.init_from_rows(init)
km.run(4)
km.init_from_rows(init); km.set_active(np.ones(m, np.uint8))
_lib.synchronize(); t0 = time.perf_counter()
it, _, _, paused = km.run(10)
_lib.synchronize(); print("run(10):", (time.perf_counter() - t0) * 1e3 / 10, "ms per iteration", it, paused)
cb = km.get_centroids()
enc = _lib.PQEncoder(cb, _lib.EUCLIDEAN)
codes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
for _ in range(3): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
_lib.synchronize(); t0 = time.perf_counter()
for _ in range(10): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
_lib.synchronize(); print("encode:", (time.perf_counter() - t0) * 1e3 / 10, "ms per pass")
