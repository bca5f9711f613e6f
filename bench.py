#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): PQ m=8, k=256, L2 on 1,000,000 x 128 f32 rows PER GPU
(weak scaling), synthetic Uniform[0,1) rows generated on the device (reference harness
distribution, src/bin/common.rs:43-53), codebooks trained by a few untimed Lloyd iterations.

A "step" is one encode pass (nearest-centroid assignment of every resident row in all m
subspaces -> one code byte per subspace) with inputs already resident in HBM.  K steps are
timed between barrier + synchronize brackets; value = rows encoded by all ranks / max time.
The same run also times Lloyd iterations (assign + update + all-reduce + finalize) and, on
rank 0 at N=1, the CPU restatement of the reference (oracle/, labelled "port").

One JSON line on stdout (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.md section 3 / SURVEY.md section 8(d)
PEAK_F32_MFMA_TFLOPS = 157.3
N_PER_GPU, DIM, M, K = 1_000_000, 128, 8, 256
DATA_SEED, TRAIN_ITERS = 66, 4


def cpu_baseline(m, k, dim, codebooks, target_seconds=12.0):
    """The reference's encode loop (src/pq.rs:167-199, driven one vector at a time on ONE
    thread by src/bin/eval_pq.rs:54-57) as restated by the oracle, timed on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O  # cpu_baseline leg only
    from vq_amd import _lib

    orc = O.get()
    probe = 2000
    X = _lib.synth_uniform_host(probe, dim, DATA_SEED, 0)
    t0 = time.perf_counter()
    orc.pq_encode(O.SQUARED_EUCLIDEAN, X, codebooks, want_f16=True, threads=1)
    per_row = (time.perf_counter() - t0) / probe
    n = int(min(200_000, max(probe, target_seconds / max(per_row, 1e-9))))
    X = _lib.synth_uniform_host(n, dim, DATA_SEED, 0)
    t0 = time.perf_counter()
    orc.pq_encode(O.SQUARED_EUCLIDEAN, X, codebooks, want_f16=True, threads=1)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "vectors/s", "cores": 1, "kind": "port",
           "sample": f"first {n} rows of the same synthetic matrix, same codebooks; single-thread "
                     "encode like the reference (src/bin/eval_pq.rs:54-57)"}
    # context: all host cores over rows (the reference does NOT do this for encode)
    nt = orc.max_threads()
    t0 = time.perf_counter()
    orc.pq_encode(O.SQUARED_EUCLIDEAN, X, codebooks, want_f16=True, threads=nt)
    out["all_cores_value"] = n / (time.perf_counter() - t0)
    out["all_cores"] = nt
    # SURVEY.md 8(d) variant B ("generous"): same source built -O3 -march=native on this host,
    # rows over all cores -- a stand-in for the reference's `simd` + rayon-over-rows best case
    try:
        fast = O.get(native=True)
        fast.pq_encode(O.SQUARED_EUCLIDEAN, X[:2000], codebooks, want_f16=True, threads=nt)
        t0 = time.perf_counter()
        fast.pq_encode(O.SQUARED_EUCLIDEAN, X, codebooks, want_f16=True, threads=nt)
        out["generous_value"] = n / (time.perf_counter() - t0)
        out["generous_note"] = f"-O3 -march=native build, {nt} threads over rows"
    except Exception as e:  # the native build is best effort (needs gcc on the bench host)
        out["generous_note"] = f"native build unavailable: {e}"
    # k-means iteration, faithful threading (SURVEY.md 8(d) variant A): assignment over all host cores
    # like rayon's par_iter (src/core/vector.rs:417-423), update serial, subspaces one after the other
    # (src/pq.rs:121), on a bounded sample of the same matrix; a Lloyd iteration is linear in the rows
    nk = min(100_000, n)
    Xk = X[:nk]
    sd = dim // m
    t0 = time.perf_counter()
    for s_ in range(m):
        orc.lloyd_step(Xk[:, s_ * sd:(s_ + 1) * sd], codebooks[s_], threads=nt)
    dtk = time.perf_counter() - t0
    out["kmeans"] = {"rows": nk, "ms_per_iter": dtk * 1e3, "threads": nt,
                     "iter_per_s_extrapolated_to_workload_rows": 1.0 / (dtk * N_PER_GPU / nk),
                     "note": "one Lloyd iteration over all m subspaces on the sample, assign parallel / update serial"}
    return out


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary (separate
    --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/run_profile.sh).  Units are KiB; FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of a coalesced stream).
    Returns (bytes or None, source)."""
    import glob

    base = kernel_name.split("<")[0]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(path)).get(base)
            if d and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                kib = 2.0 * d["FETCH_SIZE"]["avg_per_launch"] + d["WRITE_SIZE"]["avg_per_launch"]
                return kib * 1024.0, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=N_PER_GPU, help="rows per GPU (default: the workload's)")
    ap.add_argument("--kmeans-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engine", choices=["auto", "exact", "mfma", "bf16"], default="auto")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from vq_amd import _lib
    from vq_amd.sharded import Comm, HipShard, ShardedKMeans

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a MI355X (no GPU visible); there is no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    # VQ_BENCH_FORCE_COMM=1: keep the per-iteration RCCL all-reduce on for a single rank (measures what the
    # collective path adds to an iteration without a second GPU)
    force_comm = world == 1 and os.environ.get("VQ_BENCH_FORCE_COMM") == "1"
    if world > 1 or force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if force_comm:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    _lib.load()
    _lib.set_device(local_rank)
    stream = torch.cuda.Stream()
    engine = {"auto": _lib.ENGINE_AUTO, "exact": _lib.ENGINE_EXACT, "mfma": _lib.ENGINE_MFMA,
              "bf16": _lib.ENGINE_MFMA_BF16}[args.engine]

    def barrier():
        if world > 1:
            dist.barrier()

    with torch.cuda.stream(stream):
        _lib.set_stream(stream.cuda_stream)
        n = args.rows
        n_global = n * world
        ds = _lib.Dataset.synthetic(n, DIM, DATA_SEED, rank * n)

        # ---- codebooks: a few (untimed) global Lloyd iterations from strided init rows ----
        comm = Comm(force=force_comm)
        shard = HipShard(ds, M, K, rank * n, engine)
        skm = ShardedKMeans(shard, n_global, comm)
        init = np.array([[(j * (n_global // K) + s) % n_global for j in range(K)] for s in range(M)], np.int64)
        skm.init_from_global_rows(init)
        for _ in range(TRAIN_ITERS):
            counts, changed = skm.step()
        codebooks = shard.get_centroids()

        # ---- k-means iterations/s (whole job) ---------------------------------------------
        for _ in range(2):
            skm.step()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.kmeans_iters):
            skm.step()
        torch.cuda.synchronize()
        barrier()
        km_dt = time.perf_counter() - t0
        km_t = torch.tensor([km_dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(km_t, op=dist.ReduceOp.MAX)
        km_dt = float(km_t.item())

        # ---- encode: the timed region -------------------------------------------------------
        enc = _lib.PQEncoder(codebooks, _lib.SQUARED_EUCLIDEAN)
        enc.set_engine(engine)
        codes = torch.empty((n, M), dtype=torch.uint8, device="cuda")
        f16 = torch.empty((n, DIM), dtype=torch.float16, device="cuda")
        xptr = ds.device_ptr
        for _ in range(args.warmup):
            enc.encode_device(xptr, n, codes.data_ptr(), None)
        torch.cuda.synchronize()
        barrier()
        _lib.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            enc.encode_device(xptr, n, codes.data_ptr(), None)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        calls, primary_ms, recheck_ms = _lib.profile_collect()
        _lib.set_profiling(False)
        rechecked, used_engine = _lib.last_assign_stats()
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

        # ---- same pass with the reference-shaped f16 reconstruction written too -------------
        for _ in range(2):
            enc.encode_device(xptr, n, codes.data_ptr(), f16.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = max(3, args.steps // 4)
        for _ in range(reps):
            enc.encode_device(xptr, n, codes.data_ptr(), f16.data_ptr())
        torch.cuda.synchronize()
        dt_f16 = (time.perf_counter() - t0) / reps

        checksum = int(codes.to(torch.int64).sum().item())

        extras = {}
        if rank == 0 and world == 1:
            # (a) H2D-inclusive: pageable host rows in, codes back to the host (never the headline)
            nh = min(n, 250_000)
            Xh = _lib.synth_uniform_host(nh, DIM, DATA_SEED, 0)
            enc.encode(Xh, want_codes=True, want_f16=False)
            t0 = time.perf_counter()
            for _ in range(3):
                enc.encode(Xh, want_codes=True, want_f16=False)
            extras["encode_host_in_host_out_vectors_per_s"] = nh * 3 / (time.perf_counter() - t0)
            extras["encode_host_rows"] = nh
            # (b) clustered data once (mixture of K Gaussians around the trained centroids' scale):
            # uniform data is the worst case for near-ties, this is the friendly one
            g = torch.Generator(device="cuda").manual_seed(DATA_SEED)
            centers = torch.rand((K, DIM), device="cuda", generator=g)
            which = torch.randint(0, K, (n,), device="cuda", generator=g)
            Xc = (centers[which] + 0.02 * torch.randn((n, DIM), device="cuda", generator=g)).contiguous()
            torch.cuda.synchronize()
            dsc = _lib.Dataset.from_device(Xc.data_ptr(), n, DIM)
            kmc = _lib.KMeans(dsc, M, K)
            kmc.set_engine(engine)
            kmc.init_from_rows(np.array([[(j * (n // K) + s) % n for j in range(K)] for s in range(M)], np.uint64))
            for _ in range(TRAIN_ITERS):
                kmc.step()
            encc = _lib.PQEncoder(kmc.get_centroids(), _lib.SQUARED_EUCLIDEAN)
            encc.set_engine(engine)
            encc.encode_device(Xc.data_ptr(), n, codes.data_ptr(), None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                encc.encode_device(Xc.data_ptr(), n, codes.data_ptr(), None)
            torch.cuda.synchronize()
            extras["clustered_data_vectors_per_s"] = n * 5 / (time.perf_counter() - t0)
            extras["clustered_data_recheck_fraction"] = _lib.last_assign_stats()[0] / float(n * M)
            encc.close()
            kmc.close()
            dsc.close()

    if rank == 0:
        value = n_global * args.steps / dt
        flop_per_row = 2.0 * K * DIM  # the -2.x.c contraction only
        kern_s = primary_ms / 1e3 / max(calls, 1)
        achieved = flop_per_row * n / kern_s / 1e12 if kern_s > 0 else 0.0
        kernel_name = {2: "k_assign_screen<16,16>", 3: "k_assign_screen_bf16_x32<16,8>"}.get(used_engine, "k_assign_exact")
        traffic, traffic_src = pmc_traffic(kernel_name) if n == N_PER_GPU else (None, None)
        line = {
            "metric": "pq_encode_vectors_per_s",
            "value": value,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"PQ m={M} k={K} L2 encode on {n}x{DIM} f32 rows per GPU, device-resident "
                            "(BASELINE.json configs[1]); codes out (1 byte per subspace)",
                "rows_per_gpu": n, "dim": DIM, "m": M, "k": K, "sub_dim": DIM // M,
                "engine": {1: "exact", 2: "fp32_mfma_screen+exact_recheck",
                           3: "bf16x3_mfma_screen+exact_recheck"}.get(used_engine, str(used_engine)),
                "recheck_fraction": rechecked / float(n * M),
                "codebooks": f"{TRAIN_ITERS} Lloyd iterations from strided init rows",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": kernel_name,
                "note": "achieved = algorithmic 2*k*D flop per row / device time of the screen kernel; with the "
                        "bf16-split engine the contraction runs as 6 bf16 products per fp32 product on the bf16 "
                        "matrix pipe and the kernel is VALU-bound (epilogue), see DESIGN.md 4.1",
                "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                "flop_per_launch": flop_per_row * n,
                "avg_launch_ms": kern_s * 1e3,
                "recheck_avg_launch_ms": recheck_ms / max(calls, 1),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": (4.0 * DIM + M) * n,
            },
            "kmeans_iter_per_s": args.kmeans_iters / km_dt,
            "kmeans_ms_per_iter": km_dt / args.kmeans_iters * 1e3,
            "kmeans_rows_global": n_global,
            "encode_f16_out_vectors_per_s_per_gpu": n / dt_f16,
            "codes_checksum_rank0": checksum,
        }
        line.update(extras)
        r32, r16, trusted = _lib.selftest()
        line["bf16_mfma_selftest"] = {"ratio_32x32x16": r32, "ratio_16x16x32": r16, "budget": 32.0, "trusted": trusted}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(M, K, DIM, codebooks)
        print(json.dumps(line))
    if world > 1 or force_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
