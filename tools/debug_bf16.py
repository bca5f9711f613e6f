import sys, numpy as np
sys.path.insert(0, '.')
from vq_amd import _lib
rng = np.random.default_rng(0)
for (n, d, m, k) in [(64, 64, 4, 16), (512, 128, 8, 256), (512, 128, 16, 256)]:
    X = rng.random((n, d), dtype=np.float32)
    cb = rng.random((m, k, d // m), dtype=np.float32)
    res = {}
    for eng in (1, 2, 3):
        enc = _lib.PQEncoder(cb, 0); enc.set_engine(eng)
        codes, _ = enc.encode(X, want_f16=False)
        res[eng] = (codes, _lib.last_assign_stats())
        enc.close()
    bad = (res[3][0] != res[1][0])
    print((n, d, m, k), "f32 mismatches", int((res[2][0] != res[1][0]).sum()), "bf16 mismatches", int(bad.sum()), "of", bad.size,
          "rechecked f32/bf16", res[2][1][0], res[3][1][0])
    if bad.any():
        rows, subs = np.nonzero(bad)
        print(" first bad (row, sub, got, want):", [(int(r), int(s), int(res[3][0][r, s]), int(res[1][0][r, s])) for r, s in list(zip(rows, subs))[:8]])
        print(" bad per subspace", bad.sum(axis=0), " bad rows mod 16 hist", np.bincount(rows % 16, minlength=16))
