"""Host-buffer entry points at C2's shape: rows in host memory -> codes / f16 in host memory, and the data-set upload.
    python tools/host_xfer.py            (VQHIP_NO_XFER_LANES=1, VQHIP_XFER_LANES=n, VQHIP_XFER_CHUNK_MB=m for A/B)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import _lib
_lib.load(); _lib.set_device(0)
if os.environ.get("WITH_TORCH_STREAM") == "1":  # as bench.py runs it: torch initialised, the library on a torch stream
    import torch
    _ts = torch.cuda.Stream()
    _lib.set_stream(_ts.cuda_stream)
n, d, m, k = 1_000_000, 128, 8, 256
X = _lib.synth_uniform_host(n, d, 66, 0)
cb = np.random.default_rng(1).random((m, k, d // m), dtype=np.float32)
enc = _lib.PQEncoder(cb, _lib.EUCLIDEAN)
lib = _lib.load()
codes = np.empty((n, m), np.uint8); f16 = np.empty((n, d), np.uint16)


def t(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


for label, c, f in (("codes out", codes, None), ("f16 out", None, f16), ("codes + f16 out", codes, f16)):
    best, med = t(lambda: _lib.check(lib.vqhip_pq_encode(enc.raw, _lib.ptr(X, _lib._f32p), n, _lib.ptr(c, _lib._u8p), _lib.ptr(f, _lib._u16p))))
    print(f"host rows in, {label} (caller's buffers reused): best {best * 1e3:.2f} ms = {n / best:.3e} vec/s, median {med * 1e3:.2f} ms", flush=True)
best, med = t(lambda: enc.encode(X, want_codes=False, want_f16=True))
print(f"PQEncoder.encode f16 (fresh numpy output each call): best {best * 1e3:.2f} ms = {n / best:.3e} vec/s, median {med * 1e3:.2f}")
best, med = t(lambda: _lib.Dataset.from_host(X).close())
print(f"Dataset.from_host 512 MB: best {best * 1e3:.2f} ms = {X.nbytes / best / 1e9:.1f} GB/s, median {med * 1e3:.2f}")
