"""TSVQ encode with the f16 reconstruction (BASELINE configs[3]'s encode step: 4 D bytes in + 2 D + 4 out per vector), device-resident rows."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vq_amd import _lib, TSVQ, Distance
from vq_amd.tsvq import build_tree
lib = _lib.load(); _lib.set_device(0)
n, d, depth = 1_000_000, 128, 8
ds = _lib.Dataset.synthetic(n, d, 66, 0)
cent, left, right = build_tree(ds, depth)
t = TSVQ.from_tree(cent, left, right, Distance.euclidean())
leaf = torch.empty(n, dtype=torch.int32, device="cuda"); f16 = torch.empty((n, d), dtype=torch.float16, device="cuda")
def once(): _lib.check(lib.vqhip_tsvq_encode_device(t._enc.raw, C.c_void_p(ds.device_ptr), n, C.c_void_p(leaf.data_ptr()), C.c_void_p(f16.data_ptr())))
for _ in range(3): once()
_lib.synchronize(); t0 = time.perf_counter()
for _ in range(20): once()
_lib.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 20
print(f"tsvq encode + f16: {ms:.4f} ms = {(6 * d + 4) * n / ms / 1e9:.2f} TB/s")
