"""ADC one-scan schedule at large n (20M / 70M rows) against the oracle: python tools/adc_big.py"""
import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "oracle")]
import numpy as np, oracle as O
from vq_amd import _lib
F = np.float32
rng = np.random.default_rng(5)
orc = O.get()
for (n, m, sd, k, topk, nq) in [(20_000_000, 8, 4, 256, 10, 8), (20_000_000, 8, 4, 256, 200, 3), (70_000_000, 4, 2, 64, 10, 2)]:
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m), dtype=np.uint8)
    Q = rng.standard_normal((nq, m * sd)).astype(F)
    enc = _lib.PQEncoder(cb, O.SQUARED_EUCLIDEAN)
    enc.adc_set_codes(codes)
    for _ in range(2): idx, dist = enc.adc_search(None, Q, topk)
    t0 = time.perf_counter(); idx, dist = enc.adc_search(None, Q, topk); dt = time.perf_counter() - t0
    red = enc.adc_last_redone()
    wi, wd = orc.adc_search(O.SQUARED_EUCLIDEAN, cb, codes, Q, topk)
    ok = (idx == wi).all() and (dist.view(np.uint32) == wd.view(np.uint32)).all()
    print(f"n={n} m={m} k={k} topk={topk} nq={nq}: {dt*1e3:.2f} ms, redone {red}, equal {ok}", flush=True)
    enc.close()
