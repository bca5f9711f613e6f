"""Is the sporadic 30-40 ms TSVQ build (tools/tsvq_anomaly.py) deferred work of a preceding 512 MB free + allocation?
Variant A: replace the dataset, build 6 times at once.  Variant B: the same with a 0.3 s pause after the replacement.
Variant C: keep one dataset (no free / allocation between the groups)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_amd import TSVQ, Distance, _lib
from vq_amd.tsvq import build_tree
_lib.load(); _lib.set_device(0)
n, d, depth = 1_000_000, 128, 12


def group(ds):
    ts = []
    for rep in range(6):
        _lib.synchronize(); t0 = time.perf_counter()
        cent, left, right = build_tree(ds, depth)
        _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t = TSVQ.from_tree(cent, left, right, Distance("cosine"))
    t.leaf_ids(ds.read(0, 1000))
    return ts


for variant in ("A replace", "B replace + 0.3 s pause", "C keep"):
    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    group(ds)  # process warm-up
    for case in range(6):
        if not variant.startswith("C"):
            ds.close()
            ds = _lib.Dataset.synthetic(n, d, 66, 0)
            if variant.startswith("B"):
                time.sleep(0.3)
        ts = group(ds)
        print(f"{variant:26s} group {case}: " + " ".join(f"{x:6.1f}" for x in ts), flush=True)
    ds.close()
