#!/usr/bin/env python3
"""TSVQ encode time vs batch size (device-resident), to separate fixed from per-row cost."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vq_amd import TSVQ, Distance, _lib  # noqa: E402
from vq_amd.tsvq import build_tree  # noqa: E402

_lib.load()
_lib.set_device(0)
n, d, depth = 1_000_000, int(os.environ.get("D", 128)), int(os.environ.get("DEPTH", 8))
ds = _lib.Dataset.synthetic(n, d, 66, 0)
cent, left, right = build_tree(ds, depth)
t = TSVQ.from_tree(cent, left, right, Distance.euclidean())
leaf = torch.empty(n, dtype=torch.int32, device="cuda")
lib = _lib.load()
for rows in (1_000_000, 500_000, 250_000, 100_000, 20_000):
    def run():
        _lib.check(lib.vqhip_tsvq_encode_device(t._enc.raw, C.c_void_p(ds.device_ptr), rows,
                                                C.c_void_p(leaf.data_ptr()), None))
    run()
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"rows={rows:8d}  {dt*1e6:8.1f} us  {rows/dt/1e9:6.2f} Grows/s  undecided={t.last_encode_stats()[1]}")
