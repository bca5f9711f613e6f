import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import vq_amd as pyvq
from vq_amd import _lib
X = _lib.synth_uniform_host(20000, 128, 66, 0)
pq = pyvq.ProductQuantizer(X, 8, 256, max_iters=3)
lib = _lib.load()
v = np.ascontiguousarray(X[5:6]); out = np.empty((1, 128), np.uint16)
vp = v.ctypes.data_as(_lib._f32p); op = out.ctypes.data_as(_lib._u16p); raw = pq._enc.raw
def per_call(fn, reps=5000, warm=3000):
    for _ in range(warm): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
print("raw ctypes vqhip_pq_encode(1 row, f16 out): %.1f us" % per_call(lambda: lib.vqhip_pq_encode(raw, vp, 1, None, op)))
print("pq.quantize(vector): %.1f us" % per_call(lambda: pq.quantize(X[5])))
print("enc.encode(v, want_codes=False): %.1f us" % per_call(lambda: pq._enc.encode(v, want_codes=False, want_f16=True)))
