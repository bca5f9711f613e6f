"""The launch floor of this pool next to the two latency-bound paths: what ONE tiny kernel + the wait for
it costs through the same library (vqhip_dequantize_f16_device over 8 values on the library's stream, then
vqhip_synchronize), what two dependent ones cost, and what `pq.quantize(one vector)` and one C1 Lloyd iteration cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vq_amd as pyvq
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
src = torch.zeros(64, dtype=torch.float16, device="cuda"); dst = torch.zeros(64, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()


def per_call(fn, reps=3000, warm=3000):
    for _ in range(warm): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6


def one():
    _lib.dequantize_f16_device(src.data_ptr(), 8, dst.data_ptr()); _lib.synchronize()


def two():
    _lib.dequantize_f16_device(src.data_ptr(), 8, dst.data_ptr()); _lib.dequantize_f16_device(src.data_ptr(), 8, dst.data_ptr()); _lib.synchronize()


def ten():
    for _ in range(10): _lib.dequantize_f16_device(src.data_ptr(), 8, dst.data_ptr())
    _lib.synchronize()


a, b, c = per_call(one), per_call(two), per_call(ten, 1000, 1000)
print(f"one 8-element kernel + wait: {a:.1f} us;  two dependent: {b:.1f} us;  ten: {c:.1f} us  -> {(c - a) / 9:.1f} us per further dependent launch")
X = _lib.synth_uniform_host(20000, 128, 66, 0)
pq = pyvq.ProductQuantizer(X, 8, 256, max_iters=3)
t = pyvq.TSVQ(X, 8)
print(f"pq.quantize(one vector): {per_call(lambda: pq.quantize(X[5])):.1f} us per call;  tsvq.quantize: {per_call(lambda: t.quantize(X[5])):.1f} us")
n, d, m, k = 10_000, 64, 4, 16
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
for iters in (1, 10, 40):
    ts = []
    for _ in range(30):
        km.init_from_rows(init); km.set_active(np.ones(m, np.uint8)); _lib.synchronize()
        t0 = time.perf_counter(); it, _, _, _ = km.run(iters); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"C1 km.run({iters}): median {ts[len(ts) // 2]:.1f} us per call = {ts[len(ts) // 2] / iters:.1f} us per iteration (two launches each)")
