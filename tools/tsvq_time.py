"""TSVQ build timing + tree checksum.
    python tools/tsvq_time.py            four shapes on the bench's Uniform[0,1) rows
    python tools/tsvq_time.py c4         BASELINE configs[3] only (1M x 128, depth 8)
    python tools/tsvq_time.py normal     C4's shape on N(0,1) rows (zero-mean columns: the hard case of the exact column sums)
    python tools/tsvq_time.py uniform-0.5   ... on Uniform[-1/2, 1/2)
"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
arg = sys.argv[1] if len(sys.argv) > 1 else ""


def run(ds, label, depth, reps=6):
    ts = []
    for rep in range(reps):
        _lib.synchronize(); t0 = time.perf_counter()
        cent, left, right = build_tree(ds, depth)
        _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"TSVQ build {label} depth={depth}: " + " ".join(f"{x:.2f}" for x in ts) + f" ms; nodes {len(left)} crc {zlib.crc32(cent.tobytes()) & 0xffffffff:08x}", flush=True)


if arg in ("normal", "uniform-0.5"):
    n, d, depth = 1_000_000, 128, 8
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d), dtype=np.float32) if arg == "normal" else rng.random((n, d), dtype=np.float32) - np.float32(0.5)
    ds = _lib.Dataset.from_host(X)
    run(ds, f"{arg} n={n} d={d}", depth, reps=4)
    ds.close()
else:
    cases = ((1_000_000, 128, 8),) if arg == "c4" else ((1_000_000, 128, 8), (1_000_000, 128, 12), (1_000_000, 384, 5), (200_000, 768, 6))
    for (n, d, depth) in cases:
        ds = _lib.Dataset.synthetic(n, d, 66, 0)
        run(ds, f"n={n} d={d}", depth)
        ds.close()
