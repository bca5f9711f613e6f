"""Does freeing host memory that was the target of a multi-MB device->host copy delay the next kernels?
G: between builds, ds.read() of 4 MB into a fresh numpy array that is dropped at once;  H: the same, arrays kept alive;
I: 256 KB reads, dropped."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
n, d, depth = 1_000_000, 128, 12
ds = _lib.Dataset.synthetic(n, d, 66, 0)
for _ in range(3): build_tree(ds, depth)
keep = []
for variant, rows, hold in (("G 4 MB read, dropped", 8192, False), ("H 4 MB read, kept", 8192, True), ("I 256 KB read, dropped", 512, False), ("G again", 8192, False)):
    ts = []
    for rep in range(24):
        a = ds.read(0, rows)
        if hold: keep.append(a)
        del a
        _lib.synchronize(); t0 = time.perf_counter()
        build_tree(ds, depth)
        _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{variant:24s}: " + " ".join(f"{x:5.1f}" for x in ts), flush=True)
