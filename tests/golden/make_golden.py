#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/vq_oracle.c).

The reference ships no golden vectors and cannot be run here (Rust, no toolchain), so the
fixtures are oracle outputs: they pin the oracle against regressions (CPU suite) and give
the GPU suite fixed inputs/expected outputs that do not depend on the oracle being rebuilt
identically on the GPU box.  Inputs come from numpy's PCG64 with the seeds below.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

F = np.float32


def encode_case(orc, name, X, cb):
    out = {"X": X, "codebooks": cb}
    for metric, mname in ((0, "sqeuclid"), (1, "euclid"), (2, "manhattan"), (3, "cosine")):
        codes, f16 = orc.pq_encode(metric, X, cb)
        out[f"codes_{mname}"] = codes.astype(np.uint8)
        out[f"f16_{mname}"] = f16
    np.savez_compressed(os.path.join(HERE, name), **out)


def main():
    orc = O.get()
    rng = np.random.default_rng(20261002)

    # 1. encode, uniform data (reference harness distribution), config-1 shape m=4 k=16 sd=16
    X = rng.random((512, 64), dtype=F)
    cb = rng.random((4, 16, 16), dtype=F)
    encode_case(orc, "encode_uniform_m4_k16.npz", X, cb)

    # 2. encode, config-2 shape m=8 k=256 sd=16, gaussian
    X = rng.standard_normal((300, 128)).astype(F)
    cb = rng.standard_normal((8, 256, 16)).astype(F)
    encode_case(orc, "encode_normal_m8_k256.npz", X, cb)

    # 3. encode, adversarial: duplicate centroids, 1-ulp neighbours, zero rows/centroids, lattice
    X = rng.integers(0, 3, (256, 32)).astype(F)
    cb = rng.integers(0, 3, (4, 32, 8)).astype(F)
    cb[:, 7] = cb[:, 2]
    cb[0, 9] = np.nextafter(cb[0, 4], F(9))
    cb[1, 0] = 0
    X[17] = 0
    encode_case(orc, "encode_adversarial_m4_k32.npz", X, cb)

    # 4. one Lloyd step and a full fit with injected draws (no cluster empties in the step case)
    X = rng.random((2048, 32), dtype=F)
    m, k, sd = 2, 16, 16
    init = np.array([[j * 128 + s for j in range(k)] for s in range(m)], np.uint64)
    step = {"X": X, "init_rows": init}
    for s in range(m):
        c0 = X[init[s].astype(np.int64), s * sd:(s + 1) * sd]
        c1, assign, counts, changed = orc.lloyd_step(X[:, s * sd:(s + 1) * sd], c0)
        step[f"centroids_out_{s}"] = c1
        step[f"assign_{s}"] = assign.astype(np.uint8)
        step[f"counts_{s}"] = counts
        step[f"changed_{s}"] = np.array(changed)
    np.savez_compressed(os.path.join(HERE, "lloyd_step_m2_k16.npz"), **step)

    reseed = np.array([[5, 6, 7, 8, 9, 10, 11, 12]] * m, np.uint64)
    cbk, iters = orc.pq_fit(X, m, k, 10, init, reseed_rows=reseed)
    np.savez_compressed(os.path.join(HERE, "pq_fit_m2_k16.npz"), X=X, init_rows=init,
                        reseed_rows=reseed, codebooks=cbk, iters=iters)

    # 5. TSVQ: tree + leaves for the 4 metrics
    X = rng.standard_normal((600, 12)).astype(F)
    tree = orc.tsvq_build(X, 5)
    Q = rng.standard_normal((200, 12)).astype(F)
    out = {"X": X, "Q": Q, "centroids": tree["centroids"], "left": tree["left"],
           "right": tree["right"], "node_rows": tree["node_rows"]}
    for metric, mname in ((0, "sqeuclid"), (1, "euclid"), (2, "manhattan"), (3, "cosine")):
        leaf, f16 = orc.tsvq_encode(metric, Q, tree)
        out[f"leaf_{mname}"] = leaf
        out[f"f16_{mname}"] = f16
    np.savez_compressed(os.path.join(HERE, "tsvq_depth5.npz"), **out)


if __name__ == "__main__":
    main()
