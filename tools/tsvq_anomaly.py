"""VERDICT r1 'undocumented anomaly': TSVQ depth-12 build 12.4 ms vs 39.9 ms in two consecutive sweep
cases although the build does not depend on the metric.  Times every single build."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_amd import TSVQ, Distance, _lib  # noqa: E402
from vq_amd.tsvq import build_tree  # noqa: E402

_lib.load()
_lib.set_device(0)
for depth in (12, 8, 12):
    for case in range(3):
        ds = _lib.Dataset.synthetic(1_000_000, 128, 66, 0)
        ts = []
        for rep in range(5):
            _lib.synchronize()
            t0 = time.perf_counter()
            cent, left, right = build_tree(ds, depth)
            _lib.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        # an encoder between the builds, as the sweep had
        t = TSVQ.from_tree(cent, left, right, Distance("cosine" if case % 2 else "squared_euclidean"))
        t.leaf_ids(ds.read(0, 1000))
        ds.close()
        print(f"depth {depth} dataset #{case}: builds (ms) " + " ".join(f"{x:7.2f}" for x in ts), flush=True)
