import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = 1_000_000, 128, 8, 256
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
km.run(4)
enc = _lib.PQEncoder(km.get_centroids(), _lib.SQUARED_EUCLIDEAN)
codes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
for _ in range(20): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
_lib.synchronize()
out = []
for rep in range(5):
    _lib.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(50): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
    _lib.synchronize(); w = (time.perf_counter() - t0) / 50 * 1e3
    calls, prim, rech = _lib.profile_collect(); _lib.set_profiling(False)
    out.append((w, prim / calls))
print(os.environ.get("VQHIP_LIB_PATH", "default"), "encode ms/step (wall, screen kernel):", " ".join(f"{a:.4f}/{b:.4f}" for a, b in out))
for rep in range(3):
    _lib.synchronize(); t0 = time.perf_counter()
    it, _, _, _ = km.run(20)
    _lib.synchronize(); print("   run(20) ms/iter", (time.perf_counter() - t0) / 20 * 1e3)
