"""C4 TSVQ build timing (1M x 128, depth 8) + tree checksum"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
cases = ((1_000_000, 128, 8),) if len(sys.argv) > 1 and sys.argv[1] == 'c4' else ((1_000_000, 128, 8), (1_000_000, 128, 12), (1_000_000, 384, 5), (200_000, 768, 6))
for (n, d, depth) in cases:
    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    ts = []
    for rep in range(6):
        _lib.synchronize(); t0 = time.perf_counter()
        cent, left, right = build_tree(ds, depth)
        _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"TSVQ build n={n} d={d} depth={depth}: " + " ".join(f"{x:.2f}" for x in ts) + f" ms; nodes {len(left)} crc {zlib.crc32(cent.tobytes()) & 0xffffffff:08x}", flush=True)
    ds.close()
