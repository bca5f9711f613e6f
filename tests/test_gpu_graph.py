"""k-means steps replayed as a hipGraph (launch-bound sizes) give exactly the eager results:
the same fit is run in two child processes, VQHIP_GRAPH=0 (plain launches) and VQHIP_GRAPH=1
(capture on the second step, replays afterwards), including steps after subspaces converge and
after empty-cluster reseeds (which change the captured launch sequence's key or inputs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VQ_REPO"])
from vq_amd import _lib
from vq_amd.pq import fit_codebooks
_lib.load(); _lib.set_device(0)
rng = np.random.default_rng(12)
out = {}
for name, (n, d, m, k, iters) in {"small": (5000, 64, 4, 16, 12), "dupes": (3000, 32, 4, 32, 9), "c2ish": (60000, 128, 8, 256, 6)}.items():
    X = rng.random((n, d), dtype=np.float32)
    if name == "dupes":
        X[:] = X[rng.integers(0, 40, n)]          # 40 distinct rows: empty clusters -> reseeds, early convergence
    ds = _lib.Dataset.from_host(X)
    stats = {}
    cb = fit_codebooks(ds, m, k, iters, seed=5, stats=stats)
    out[name + "_cb"] = cb
    out[name + "_iters"] = stats["iters"]
    km = _lib.KMeans(ds, m, k)                     # manual stepping: counts / changed / assignments per step
    km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
    for it in range(4):
        counts, changed = km.step()
        out[f"{name}_counts{it}"] = counts
        out[f"{name}_changed{it}"] = changed
    out[name + "_assign"] = km.get_assignments()
    out[name + "_rechecked"] = np.array(_lib.last_assign_stats())
    km.close(); ds.close()
np.savez(sys.argv[1], **out)
print("WORKER_OK")
'''


def _run(tmp_path, graph):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = tmp_path / f"g{graph}.npz"
    env = dict(os.environ, VQHIP_GRAPH=str(graph), VQ_REPO=ROOT)
    r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def test_graph_replay_equals_eager(tmp_path):
    eager, graph = _run(tmp_path, 0), _run(tmp_path, 1)
    assert set(eager.files) == set(graph.files)
    for key in eager.files:
        np.testing.assert_array_equal(eager[key], graph[key], err_msg=key)
