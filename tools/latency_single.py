#!/usr/bin/env python3
"""Per-call latency of the reference-shaped single-vector API (quantize one vector per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vq_amd as pyvq
from vq_amd import _lib
X = _lib.synth_uniform_host(20000, 128, 66, 0)
pq = pyvq.ProductQuantizer(X, 8, 256, max_iters=3)
t = pyvq.TSVQ(X, 8)
for name, q in (("pq.quantize", pq), ("tsvq.quantize", t)):
    t0 = time.perf_counter()
    for i in range(500):
        q.quantize(X[i])
    cold = (time.perf_counter() - t0) / 500 * 1e6
    for i in range(3000):  # let the clocks ramp: back-to-back tiny kernels start at the idle sclk
        q.quantize(X[i])
    t0 = time.perf_counter()
    for i in range(2000):
        q.quantize(X[i])
    print(f"{name}: {(time.perf_counter()-t0)/2000*1e6:.1f} us per call sustained ({cold:.1f} us over the first 500 calls)")
b = X[:64].copy()
t0 = time.perf_counter()
for i in range(200):
    pq.quantize_batch(b)
print(f"pq.quantize_batch(64 rows): {(time.perf_counter()-t0)/200*1e6:.1f} us per call")
