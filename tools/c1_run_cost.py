"""C1-sized k-means (10k x 64, m = 4, k = 16): wall time of vqhip_kmeans_run by the number of iterations -> fixed cost per call
and cost per iteration"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = 10_000, 64, 4, 16
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
def restart():
    km.init_from_rows(init); km.set_active(np.ones(m, np.uint8))
for iters in (1, 2, 5, 10, 20, 40):
    ts = []
    for rep in range(30):
        restart(); _lib.synchronize()
        t0 = time.perf_counter()
        it, counts, ch, paused = km.run(iters)
        _lib.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"run({iters:2d}): median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  -> {ts[len(ts)//2]/max(1,int(np.max(it))):6.1f} us per iteration (ran {int(np.max(it))})")
