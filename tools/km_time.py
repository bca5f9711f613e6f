"""time one k-means iteration at C2 with HIP-event-free wall clock (several reps), for A/B of library builds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = (int(x) for x in os.environ.get("VQ_KM_SHAPE", "1000000,128,8,256").split(","))  # C3: 1000000,768,96,256
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
for _ in range(4): km.step()
ts = []
for rep in range(5):
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(20): km.step()
    _lib.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(os.environ.get("VQHIP_LIB_PATH", "default"), "step loop: kmeans ms/iter", " ".join(f"{x:.4f}" for x in ts), "min", min(ts))
ts = []
for rep in range(5):
    _lib.synchronize(); t0 = time.perf_counter()
    it, _, _, paused = km.run(20)
    _lib.synchronize(); ts.append((time.perf_counter() - t0) / max(1, int(it.max())) * 1e3)
print("device-driven run(20): kmeans ms/iter", " ".join(f"{x:.4f}" for x in ts), "min", min(ts), "paused", paused, "iters", it)
