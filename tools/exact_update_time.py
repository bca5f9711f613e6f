"""k-means with exact_update at C2 (centroids bit-identical to the reference's sequential sums): ms per iteration, for rocprofv3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = (int(x) for x in os.environ.get("VQ_KM_SHAPE", "1000000,128,8,256").split(","))
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
km.set_exact_update(True)
init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
km.init_from_rows(init)
km.run(2)
for rep in range(3):
    km.init_from_rows(init); km.set_active(np.ones(m, np.uint8)); _lib.synchronize()
    t0 = time.perf_counter(); it, _, _, _ = km.run(5); _lib.synchronize()
    print(f"exact_update run(5): {(time.perf_counter() - t0) / max(1, int(it.max())) * 1e3:.4f} ms per iteration")
