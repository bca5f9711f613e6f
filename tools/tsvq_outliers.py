"""Sporadic slow TSVQ builds: 60 depth-12 builds with the host timeline of every build above 15 ms
(VQHIP_TSVQ_TIMING, printed by the library)."""
import os, sys, time
os.environ.setdefault("VQHIP_TSVQ_TIMING", "15")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_amd import _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ds = _lib.Dataset.synthetic(1_000_000, 128, 66, 0)
ts = []
for rep in range(60):
    _lib.synchronize(); t0 = time.perf_counter()
    build_tree(ds, depth)
    _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"depth {depth}: " + " ".join(f"{x:.1f}" for x in ts), flush=True)
