#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configurations (bench.py stays the
headline contract).  One JSON line per configuration on one MI355X:

  C1  PQ m=4  k=16  Euclidean, 10k x 64     (the reference's own CPU-sized case)
  C2  PQ m=8  k=256 L2,        1M x 128     fit (10 Lloyd iterations) + encode
  C3  PQ m=96 k=256 cosine,    1M x 768     fit + cosine encode (bf16 MFMA cosine screen + exact re-check)
  C4  TSVQ depth 8 L2,         1M x 128     build + encode
  C5  PQ m=16 k=256 L2,        per-GPU shard of 100M x 128 (12.5M rows) fit iteration + encode
  C2_manhattan  C2 with Distance::Manhattan for the encode (no contraction form: exact VALU engine)
  ADC code-based top-10 search of 64 queries over the C2 codes (1M x 8 bytes)
  E   PQ m=16 k=256 Euclidean, 1M x 384     (the reference's `make eval` shape, src/bin/common.rs:10-15: sub_dim 24)

    python bench_configs.py [C1 C2 ...]
"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _sync():
    from vq_amd import _lib

    _lib.synchronize()


def pq_config(name, n, d, m, k, metric_name, iters=10, encode_reps=5, engine=None):
    import torch

    from vq_amd import _lib

    metric = {"l2": _lib.SQUARED_EUCLIDEAN, "euclidean": _lib.EUCLIDEAN, "cosine": _lib.COSINE,
              "manhattan": _lib.MANHATTAN}[metric_name]
    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    km = _lib.KMeans(ds, m, k)
    if engine is not None:
        km.set_engine(engine)
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    km.init_from_rows(init)
    km.step()  # warm-up (allocations, code objects)
    km.init_from_rows(init)
    _sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        km.step()
    _sync()
    fit_s = time.perf_counter() - t0
    cb = km.get_centroids()
    km.close()
    enc = _lib.PQEncoder(cb, metric)
    if engine is not None:
        enc.set_engine(engine)
    codes = torch.empty((n, m * (1 if k <= 256 else 2)), dtype=torch.uint8, device="cuda")  # u16 codes above 256
    f16 = torch.empty((n, d), dtype=torch.float16, device="cuda")
    enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
    _sync()
    _lib.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(encode_reps):
        enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
    _sync()
    enc_s = (time.perf_counter() - t0) / encode_reps
    calls, prim_ms, re_ms = _lib.profile_collect()
    _lib.set_profiling(False)
    rechecked, engine = _lib.last_assign_stats()
    t0 = time.perf_counter()
    for _ in range(encode_reps):
        enc.encode_device(ds.device_ptr, n, codes.data_ptr(), f16.data_ptr())
    _sync()
    enc16_s = (time.perf_counter() - t0) / encode_reps
    flop = 2.0 * k * d * n
    out = {
        "config": name, "n": n, "d": d, "m": m, "k": k, "metric": metric_name,
        "kmeans_ms_per_iter": fit_s / iters * 1e3, "kmeans_iter_per_s": iters / fit_s,
        "encode_vectors_per_s": n / enc_s, "encode_ms": enc_s * 1e3,
        "encode_f16_out_vectors_per_s": n / enc16_s,
        "engine": {1: "exact", 2: "fp32_mfma_screen+exact_recheck", 3: "bf16x3_mfma_screen+exact_recheck"}[engine],
        "recheck_fraction": rechecked / float(n * m),
        "primary_kernel_ms": prim_ms / max(calls, 1), "recheck_kernel_ms": re_ms / max(calls, 1),
        "algorithmic_tflops": flop / enc_s / 1e12,
    }
    enc.close()
    ds.close()
    return out


def tsvq_config(name, n, d, depth, reps=3):
    import torch

    from vq_amd import TSVQ, Distance, _lib
    from vq_amd.tsvq import build_tree

    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    build_tree(ds, depth)  # warm-up
    _sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        cent, left, right = build_tree(ds, depth)
    _sync()
    build_s = (time.perf_counter() - t0) / reps
    t = TSVQ.from_tree(cent, left, right, Distance.euclidean())
    leaf = torch.empty(n, dtype=torch.int32, device="cuda")
    f16 = torch.empty((n, d), dtype=torch.float16, device="cuda")
    lib = _lib.load()
    import ctypes as C

    def run():
        _lib.check(lib.vqhip_tsvq_encode_device(t._enc.raw, C.c_void_p(ds.device_ptr), n,
                                                C.c_void_p(leaf.data_ptr()), C.c_void_p(f16.data_ptr())))
    run()
    _sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    _sync()
    enc_s = (time.perf_counter() - t0) / reps
    levels = depth
    build_bytes = 4.0 * n * d * (2 * levels + 1)
    out = {
        "config": name, "n": n, "d": d, "depth": depth, "nodes": int(cent.shape[0]),
        "build_ms": build_s * 1e3, "build_algorithmic_GBps": build_bytes / build_s / 1e9,
        "encode_vectors_per_s": n / enc_s, "encode_ms": enc_s * 1e3,
        "encode_algorithmic_GBps": (4.0 * d + 2.0 * d) * n / enc_s / 1e9,
    }
    ds.close()
    return out


def adc_config(name, n, d, m, k, nq, topk, reps=3):
    """code-based search: device-resident codes, host queries, top-k out"""
    import torch

    from vq_amd import _lib

    ds = _lib.Dataset.synthetic(n, d, 66, 0)
    km = _lib.KMeans(ds, m, k)
    km.init_from_rows(np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64))
    for _ in range(3):
        km.step()
    cb = km.get_centroids()
    km.close()
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    codes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
    enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
    Q = _lib.synth_uniform_host(nq, d, 67, 0)
    enc.adc_search((codes.data_ptr(), n), Q, topk)
    _sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        idx, dist = enc.adc_search((codes.data_ptr(), n), Q, topk)
    dt = (time.perf_counter() - t0) / reps
    out = {"config": name, "n": n, "d": d, "m": m, "k": k, "queries": nq, "topk": topk, "search_ms": dt * 1e3,
           "queries_per_s": nq / dt, "code_rows_scanned_per_s": nq * n / dt,
           "scan_GBps_codes_plus_distances": nq * n * (m / 8.0 + 4.0 * 6) / dt / 1e9}
    enc.close()
    ds.close()
    return out


CONFIGS = {
    "C1": lambda: pq_config("C1", 10_000, 64, 4, 16, "euclidean"),
    "C2": lambda: pq_config("C2", 1_000_000, 128, 8, 256, "l2"),
    "C3": lambda: pq_config("C3", 1_000_000, 768, 96, 256, "cosine", iters=5, encode_reps=2),
    "C4": lambda: tsvq_config("C4", 1_000_000, 128, 8),
    "C5": lambda: pq_config("C5_per_gpu_shard", 12_500_000, 128, 16, 256, "l2", iters=3, encode_reps=2),
    "C2_manhattan": lambda: pq_config("C2_manhattan", 1_000_000, 128, 8, 256, "manhattan", iters=3, encode_reps=3),
    "SD32": lambda: pq_config("SD32_1Mx128_m4", 1_000_000, 128, 4, 256, "l2", iters=3, encode_reps=5),
    "SD48": lambda: pq_config("SD48_1Mx384_m8", 1_000_000, 384, 8, 256, "l2", iters=3, encode_reps=3),
    "SD64": lambda: pq_config("SD64_1Mx128_m2", 1_000_000, 128, 2, 256, "l2", iters=3, encode_reps=5),
    "C2_k128": lambda: pq_config("C2_k128", 1_000_000, 128, 8, 128, "l2", iters=3, encode_reps=5),
    "C2_k1024": lambda: pq_config("C2_k1024", 1_000_000, 128, 8, 1024, "l2", iters=3, encode_reps=3),
    "C2_k1024_exact": lambda: pq_config("C2_k1024_exact", 1_000_000, 128, 8, 1024, "l2", iters=3, encode_reps=3, engine=1),
    "C2_k4096": lambda: pq_config("C2_k4096", 1_000_000, 128, 8, 4096, "l2", iters=2, encode_reps=2),
    # shapes off the BASELINE list that exercise the padded / wide screens and the wider TSVQ descents
    "W100": lambda: pq_config("W100_1Mx100_m10", 1_000_000, 100, 10, 256, "l2", iters=3, encode_reps=3),
    "W300": lambda: pq_config("W300_1Mx300_m10", 1_000_000, 300, 10, 256, "l2", iters=3, encode_reps=3),
    "LBG128": lambda: pq_config("LBG128_1Mx128_m1", 1_000_000, 128, 1, 256, "l2", iters=3, encode_reps=3),
    "T384": lambda: tsvq_config("T384_eval_shape_depth5", 1_000_000, 384, 5),
    "ADC": lambda: adc_config("ADC_C2", 1_000_000, 128, 8, 256, 64, 10),
    "E": lambda: pq_config("E_eval_shape", 1_000_000, 384, 16, 256, "euclidean", iters=5, encode_reps=3),
}

if __name__ == "__main__":
    from vq_amd import _lib

    _lib.load()
    _lib.set_device(0)
    which = sys.argv[1:] or list(CONFIGS)
    for c in which:
        print(json.dumps(CONFIGS[c]()), flush=True)
