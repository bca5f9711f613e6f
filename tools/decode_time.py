"""decode / dequantize / encode-with-f16 device times at 1M x 128, m = 8 (bench.py's `decode` block and the f16-out encode)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k = 1_000_000, 128, 8, 256
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
km.run(3)
enc = _lib.PQEncoder(km.get_centroids(), _lib.SQUARED_EUCLIDEAN)
codes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
f16 = torch.empty((n, d), dtype=torch.float16, device="cuda")
out = torch.empty((n, d), dtype=torch.float32, device="cuda")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    _lib.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(f"encode codes only : {timeit(lambda: enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)):.4f} ms")
print(f"encode codes + f16: {timeit(lambda: enc.encode_device(ds.device_ptr, n, codes.data_ptr(), f16.data_ptr())):.4f} ms")
print(f"decode            : {timeit(lambda: enc.decode_device(codes.data_ptr(), n, out.data_ptr())):.4f} ms  ({520e6 / 1e9:.3f} GB)")
print(f"dequantize f16    : {timeit(lambda: _lib.dequantize_f16_device(f16.data_ptr(), n * d, out.data_ptr())):.4f} ms  ({768e6 / 1e9:.3f} GB)")
