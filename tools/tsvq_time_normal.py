"""TSVQ build on zero-mean data (the sums are random walks: the hard case for the binade guess)"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
n, d, depth = 1_000_000, 128, 8
rng = np.random.default_rng(5)
for kind in (sys.argv[1:] or ["normal", "uniform-0.5", "uniform"]):
    X = rng.standard_normal((n, d), dtype=np.float32) if kind == "normal" else rng.random((n, d), dtype=np.float32) - (np.float32(0.5) if kind == "uniform-0.5" else np.float32(0))
    ds = _lib.Dataset.from_host(X)
    ts = []
    for rep in range(4):
        _lib.synchronize(); t0 = time.perf_counter()
        cent, left, right = build_tree(ds, depth)
        _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"TSVQ build {kind} n={n} d={d} depth={depth}: " + " ".join(f"{x:.2f}" for x in ts) + f" ms; nodes {len(left)} crc {zlib.crc32(cent.tobytes()) & 0xffffffff:08x}", flush=True)
    ds.close()
