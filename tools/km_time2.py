import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = 1_000_000, 128, 8, 256
ds = _lib.Dataset.synthetic(n, d, 66, 0)
for trial in range(3):
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
    km.run(4); km.run(2)
    out = []
    for rep in range(6):
        _lib.synchronize(); t0 = time.perf_counter()
        it, _, _, paused = km.run(10)
        _lib.synchronize(); out.append((time.perf_counter() - t0) / max(1, int(it.max())) * 1e3)
        r, e = _lib.last_assign_stats()
        out.append(r)
    print("fresh fit, successive run(10) ms/iter and rechecked rows:", " ".join(f"{x:.4f}" if isinstance(x, float) else str(x) for x in out), flush=True)
    km.close()
