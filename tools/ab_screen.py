"""A/B of two library builds on ONE box (boxes of the pool differ by +-4 %): encode and k-means times at C2 (1M x 128,
m = 8, sub_dim 16) and C3's shape (1M x 768, m = 96, sub_dim 8, cosine).
    VQHIP_LIB_PATH=ab/libvqhip_base.so python tools/ab_screen.py ; python tools/ab_screen.py    (alternate a few times)"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
tag = os.path.basename(os.environ.get("VQHIP_LIB_PATH", "new"))


def run(label, n, d, m, k, metric, reps=6, inner=25):
    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
    for _ in range(2): km.step()
    cb = km.get_centroids()
    enc = _lib.PQEncoder(cb, metric)
    dcodes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(12 * inner): enc.encode_device(ds.device_ptr, n, dcodes.data_ptr(), None)   # clocks up (the first ~30 passes run up to 10 % slow)
    _lib.synchronize()
    te = []
    for _ in range(reps):
        _lib.synchronize(); t0 = time.perf_counter()
        for _ in range(inner): enc.encode_device(ds.device_ptr, n, dcodes.data_ptr(), None)
        _lib.synchronize(); te.append((time.perf_counter() - t0) / inner * 1e3)
    codes = dcodes.cpu().numpy()
    rech, eng = _lib.last_assign_stats()
    tk = []
    for _ in range(reps):
        _lib.synchronize(); t0 = time.perf_counter()
        for _ in range(inner): km.step()
        _lib.synchronize(); tk.append((time.perf_counter() - t0) / inner * 1e3)
    crc = zlib.crc32(np.ascontiguousarray(codes).tobytes()) & 0xffffffff if codes is not None else 0
    te_s = sorted(te)
    print(f"{tag:20s} {label}: encode ms min {min(te):.4f} median {te_s[len(te_s) // 2]:.4f} [{' '.join(f'{x:.4f}' for x in te)}]  kmeans ms/iter min {min(tk):.4f}  rechecked {rech} crc {crc:08x}", flush=True)
    enc.close(); km.close(); ds.close()


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "c2"): run("C2 1Mx128 m8 sd16 L2", 1_000_000, 128, 8, 256, 0)
if which in ("all", "c3"): run("C3 1Mx768 m96 sd8 cos", 1_000_000, 768, 96, 256, 3, reps=4, inner=6)
