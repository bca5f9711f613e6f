"""TSVQ encode (leaf ids, device-resident rows) timing per metric:  python tools/tsvq_enc_time.py [n d depth]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vq_amd import _lib, TSVQ, Distance
from vq_amd.tsvq import build_tree
lib = _lib.load(); _lib.set_device(0)
cases = [(1_000_000, 384, 5), (1_000_000, 128, 8)]
if len(sys.argv) == 4:
    cases = [tuple(int(x) for x in sys.argv[1:4])]
for (n, d, depth) in cases:
    ds = _lib.Dataset.synthetic(n, d, 67, 0)
    cent, left, right = build_tree(ds, depth)
    leaf = torch.empty(n, dtype=torch.int32, device="cuda")
    for name in ("squared_euclidean", "cosine", "manhattan"):
        t = TSVQ.from_tree(cent, left, right, Distance(name))
        ts = []
        for rep in range(8):
            _lib.synchronize(); t0 = time.perf_counter()
            _lib.check(lib.vqhip_tsvq_encode_device(t._enc.raw, C.c_void_p(ds.device_ptr), n, C.c_void_p(leaf.data_ptr()), None))
            _lib.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"TSVQ encode n={n} d={d} depth={depth} {name}: " + " ".join(f"{x:.3f}" for x in ts) + f" ms; stats {t.last_encode_stats()}", flush=True)
    ds.close()
