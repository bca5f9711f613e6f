"""ADC search timing: 64 / 8 queries over 1M x 8 device-resident codes (bench.py's `adc` block), for rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
n, d, m, k, topk = 1_000_000, 128, 8, 256, 10
ds = _lib.Dataset.synthetic(n, d, 66, 0)
km = _lib.KMeans(ds, m, k)
km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
km.run(3)
enc = _lib.PQEncoder(km.get_centroids(), _lib.SQUARED_EUCLIDEAN)
codes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
_lib.synchronize()
Q = _lib.synth_uniform_host(64, d, 67, 0)
for nq in ([int(x) for x in sys.argv[1:]] or (64, 8, 1)):
    for _ in range(3):
        enc.adc_search((codes.data_ptr(), n), Q[:nq], topk)
    t0 = time.perf_counter()
    for _ in range(20):
        enc.adc_search((codes.data_ptr(), n), Q[:nq], topk)
    print(f"adc {nq} queries: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
