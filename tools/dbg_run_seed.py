import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from vq_amd import _lib
import test_gpu_fuzz as T
F = np.float32
seed = int(sys.argv[1])
rng = np.random.default_rng(9000 + seed)
sd = int(rng.choice([4, 8, 12, 16, 24, 32, 10, 7])); m = int(rng.integers(1, 9)); k = int(rng.choice([2, 5, 16, 64, 100, 256]))
n = int(rng.integers(max(2 * k, 300), 30_000)); d = m * sd
kind = T.KINDS[int(rng.integers(0, len(T.KINDS)))]
X = T._draw_data(rng, n, d, kind)
if rng.random() < 0.5: X = (np.round(X * 4) / 4).astype(F)
init = np.stack([rng.choice(n, k, replace=False) for _ in range(m)]).astype(np.uint64)
for _ in range(int(rng.integers(0, 4))):
    s_, a_, b_ = int(rng.integers(0, m)), int(rng.integers(0, k)), int(rng.integers(0, k))
    if a_ != b_: X[init[s_, a_]] = X[init[s_, b_]]
_lib.load(); _lib.set_device(0)
ds = _lib.Dataset.from_host(X)
reseed = [[int(x) for x in rng.integers(0, n, 4096)] for _ in range(m)]
max_iters = int(rng.integers(1, 25))
print("n", n, "m", m, "k", k, "sd", sd, kind, "max_iters", max_iters)
def fit(use_run):
    km = _lib.KMeans(ds, m, k); km.init_from_rows(init)
    active = np.ones(m, bool); iters = np.zeros(m, np.int64); its = [iter(r) for r in reseed]
    pauses, done = 0, 0; log = []
    while done < max_iters and active.any():
        if use_run:
            it, counts, changed, paused = km.run(max_iters - done); iters += it; done += max(1, int(it.max()))
        else:
            counts, changed = km.step(); iters[active] += 1; done += 1
            paused = bool(((counts == 0) & active[:, None]).any())
        log.append((done, bool(paused), int((counts == 0).sum()), changed.astype(int).tolist()))
        if paused:
            pauses += 1
            for s, j in np.argwhere((counts == 0) & active[:, None]): km.patch_from_row(int(s), int(j), next(its[s]))
        active &= changed.astype(bool); km.set_active(active)
    cb = km.get_centroids(); km.close(); return cb, iters, pauses, log
a = fit(True); b = fit(False)
print("run :", a[1], a[2], a[3])
print("step:", b[1], b[2], b[3])
print("cb equal", a[0].tobytes() == b[0].tobytes(), "max diff", np.abs(a[0] - b[0]).max())
c = fit(False); e = fit(True)
print("step vs step equal", b[0].tobytes() == c[0].tobytes(), "run vs run equal", a[0].tobytes() == e[0].tobytes())
os.environ["VQHIP_GRAPH"] = "0"
