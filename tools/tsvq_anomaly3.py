"""What triggers the 10-30 ms late start of a build's first kernel?  D: results kept alive (nothing is unmapped);
E: results kept, but a 4 MB numpy array is allocated, touched and dropped between builds (mmap + munmap of memory the
GPU never saw); F: like E with a 64 KB array (heap, no munmap)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
n, d, depth = 1_000_000, 128, 12
ds = _lib.Dataset.synthetic(n, d, 66, 0)
keep = []
for variant, nbytes in (("D keep results", 0), ("E + 4 MB alloc/free", 4 << 20), ("F + 64 KB alloc/free", 64 << 10), ("D again", 0)):
    ts = []
    for rep in range(24):
        if nbytes:
            a = np.empty(nbytes, np.uint8); a[::4096] = 1; del a
        _lib.synchronize(); t0 = time.perf_counter()
        keep.append(build_tree(ds, depth))
        _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{variant:24s}: " + " ".join(f"{x:5.1f}" for x in ts), flush=True)
