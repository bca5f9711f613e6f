"""Evaluation report in the shape of the reference's `eval_pq` / `eval_tsvq` binaries
(src/bin/eval_pq.rs:31-72, src/bin/eval_tsvq.rs:25-59, src/bin/common.rs:9-131):

    python -m vq_amd.evalcli pq   [--seed 66 --dim 384 --m 16 --k 256 --max-iters 10]
    python -m vq_amd.evalcli tsvq [--seed 66 --dim 384 --max-depth 5]

For every sample count of `NUM_SAMPLES` it prints the reference's three lines -- training time,
quantization time (host matrix in, f16 matrix out: what `quantize` per vector produces there)
and the mean squared reconstruction error -- and the two `BenchmarkResult` fields the binaries
compute nowhere: recall@k with `calculate_recall`'s windowed protocol (common.rs:91-130) and
the memory reduction ratio.  `--json` emits one `BenchmarkResult`-shaped object per line.

Data: i.i.d. Uniform[0,1) like common.rs:43-53, from the library's counter-based generator
(the reference's StdRng stream is not reproducible outside Rust).
"""
from __future__ import annotations

import argparse
import json
import sys
import time

import numpy as np

SEED = 66                                                    # common.rs:9
NUM_SAMPLES = [1_000, 5_000, 10_000, 50_000, 100_000, 1_000_000]  # common.rs:10
DIM, M, K, MAX_ITERS = 384, 16, 256, 10                      # common.rs:12-15


def reconstruction_error(original: np.ndarray, reconstructed: np.ndarray) -> float:
    """common.rs:61-78: sum of squared differences / number of elements"""
    diff = original.astype(np.float32) - reconstructed.astype(np.float32)
    return float((diff * diff).sum(dtype=np.float64) / original.size)


def recall_at_k(original: np.ndarray, approx: np.ndarray, k: int = 10) -> float:
    """common.rs:91-130: <= 1000 strided queries; neighbours searched inside a 5000-row window
    around the query (the whole set when n <= 10 000); true neighbours by Euclidean distance on
    the originals, approximate ones on the reconstructions; ties keep index order (stable sort)."""
    n = original.shape[0]
    eval_samples = min(n, 1000)
    step = max(n // eval_samples, 1)
    window = 5000 if n > 10_000 else n
    total = 0.0
    for i in range(0, n, step):
        lo, hi = max(i - window // 2, 0), min(i + window // 2, n)
        idx = np.arange(lo, hi)
        idx = idx[idx != i]
        d_true = ((original[idx] - original[i]) ** 2).sum(axis=1)
        d_appr = ((approx[idx] - approx[i]) ** 2).sum(axis=1)
        t = idx[np.argsort(d_true, kind="stable")[:k]]
        a = idx[np.argsort(d_appr, kind="stable")[:k]]
        total += len(np.intersect1d(t, a)) / k
    return total / (n // step)


def _report(title, make_quantizer, args, code_bytes_per_vector):
    from . import _lib

    print(title)
    print("=" * len(title))
    for n in args.samples:
        X = _lib.synth_uniform_host(n, args.dim, args.seed, 0)
        t0 = time.perf_counter()
        q = make_quantizer(X)
        train_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        f16 = q.quantize_batch(X)
        quant_ms = (time.perf_counter() - t0) * 1e3
        rec = f16.astype(np.float32)
        err = reconstruction_error(X, rec)
        res = {"n_samples": n, "n_dims": args.dim, "training_time_ms": train_ms,
               "quantization_time_ms": quant_ms, "reconstruction_error": err}
        if args.recall_k > 0:
            res["recall"] = recall_at_k(X, rec, args.recall_k)
        # the reference keeps the f16 reconstruction (2 bytes per dimension); the codes are smaller
        res["memory_reduction_ratio"] = 4.0 * args.dim / (2.0 * args.dim)
        res["memory_reduction_ratio_codes"] = 4.0 * args.dim / code_bytes_per_vector(q)
        if args.json:
            print(json.dumps(res))
            continue
        print(f"\nSamples: {n}")
        print(f"  Training time: {train_ms:.0f} ms")
        print(f"  Quantization time: {quant_ms:.0f} ms")
        print(f"  Reconstruction error: {err:.6f}")
        if "recall" in res:
            print(f"  Recall@{args.recall_k}: {res['recall']:.4f}")
        print(f"  Memory reduction: {res['memory_reduction_ratio']:.1f}x as f16, "
              f"{res['memory_reduction_ratio_codes']:.1f}x as codes")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="vq_amd.evalcli")
    sub = ap.add_subparsers(dest="alg", required=True)
    for name in ("pq", "tsvq"):
        p = sub.add_parser(name)
        p.add_argument("--seed", type=int, default=SEED)
        p.add_argument("--dim", type=int, default=DIM)
        p.add_argument("--samples", type=int, nargs="+", default=NUM_SAMPLES)
        p.add_argument("--recall-k", type=int, default=10, help="0 skips the recall estimate")
        p.add_argument("--json", action="store_true")
        if name == "pq":
            p.add_argument("--m", type=int, default=M)
            p.add_argument("--k", type=int, default=K)
            p.add_argument("--max-iters", type=int, default=MAX_ITERS)
        else:
            p.add_argument("--max-depth", type=int, default=5)
    args = ap.parse_args(argv)
    from . import TSVQ, Distance, ProductQuantizer

    if args.alg == "pq":
        _report("Product Quantizer Evaluation",
                lambda X: ProductQuantizer(X, args.m, args.k, args.max_iters, Distance.euclidean(), args.seed),
                args, lambda q: q.num_subspaces)
    else:
        _report("TSVQ Evaluation", lambda X: TSVQ(X, args.max_depth, Distance.euclidean()), args,
                lambda q: max(1, (int(q.tree[0].shape[0]).bit_length() + 7) // 8))
    return 0


if __name__ == "__main__":
    sys.exit(main())
