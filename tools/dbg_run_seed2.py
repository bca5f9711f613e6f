import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from vq_amd import _lib
import test_gpu_fuzz as T
F = np.float32
seed = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(9000 + seed)
sd = int(rng.choice([4, 8, 12, 16, 24, 32, 10, 7])); m = int(rng.integers(1, 9)); k = int(rng.choice([2, 5, 16, 64, 100, 256]))
n = int(rng.integers(max(2 * k, 300), 30_000)); d = m * sd
kind = T.KINDS[int(rng.integers(0, len(T.KINDS)))]
X = T._draw_data(rng, n, d, kind)
if rng.random() < 0.5: X = (np.round(X * 4) / 4).astype(F); print("grid")
init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
for _ in range(int(rng.integers(0, 4))):
    s_, a_, b_ = int(rng.integers(0, m)), int(rng.integers(0, k)), int(rng.integers(0, k))
    if a_ != b_: X[init[s_, a_]] = X[init[s_, b_]]
_lib.load(); _lib.set_device(0)
ds = _lib.Dataset.from_host(X)
outs = []
for rep in range(6):
    km = _lib.KMeans(ds, m, k); km.init_from_rows(init)
    if len(sys.argv) > 3: km.set_engine(int(sys.argv[3]))
    for _ in range(steps): counts, changed = km.step()
    outs.append((km.get_centroids().copy(), km.get_assignments().copy(), counts.copy()))
    km.close()
for rep in range(1, 6):
    a, b = outs[0], outs[rep]
    print(rep, "centroids equal", a[0].tobytes() == b[0].tobytes(), "codes equal", (a[1] == b[1]).all(), "counts equal", (a[2] == b[2]).all(),
          "n differing centroid comps", int((a[0].view(np.uint32) != b[0].view(np.uint32)).sum()))
