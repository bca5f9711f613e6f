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
# (bench.py's f16-out figure takes 2 warm-up calls and max(3, steps // 4) timed ones: does the short loop read differently?)
for reps in (5, 5, 20, 50):
    for _ in range(2): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), f16.data_ptr())
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), f16.data_ptr())
    _lib.synchronize(); print(f"encode codes + f16, {reps:2d} timed calls: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms")
import torch as _t
_t.cuda.synchronize(); time.sleep(0.5)
for reps in (5, 20):
    for _ in range(2): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), f16.data_ptr())
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): enc.encode_device(ds.device_ptr, n, codes.data_ptr(), f16.data_ptr())
    _lib.synchronize(); print(f"after a 0.5 s pause, {reps:2d} timed calls: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms")
