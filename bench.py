#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): PQ m=8, k=256, L2 on 1,000,000 x 128 f32 rows PER GPU
(weak scaling), synthetic Uniform[0,1) rows generated on the device (reference harness
distribution, src/bin/common.rs:43-53), codebooks trained by a few untimed Lloyd iterations.

A "step" is one encode pass (nearest-centroid assignment of every resident row in all m
subspaces -> one code byte per subspace) with inputs already resident in HBM.  K steps are
timed between barrier + synchronize brackets; value = rows encoded by all ranks / max time.
The same run also times Lloyd iterations (assign + fused update + all-reduce + finalize, the
loop's decisions on the device: vqhip_kmeans_run[_sharded]) and, on rank 0 at N=1, the other
BASELINE configurations (C1, C3, C4: a `configs` block, each with its own roofline) and the
CPU restatement of the reference (oracle/, labelled "port").

    --config C5        BASELINE configs[4]'s per-GPU share: 12.5M rows per GPU, m=16 (the 8-GPU job)
    --collective torch the all-reduce through torch.distributed instead of below the C ABI

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
PEAK_HBM_GBS = 8000.0
N_PER_GPU, DIM, M, K = 1_000_000, 128, 8, 256
DATA_SEED, TRAIN_ITERS = 66, 4
WORKLOADS = {  # BASELINE.json configs[1] and configs[4] (per-GPU share of the 100M x 128 job on 8 GPUs)
    "C2": dict(rows=1_000_000, dim=128, m=8, k=256, label="BASELINE.json configs[1]"),
    "C5": dict(rows=12_500_000, dim=128, m=16, k=256, label="BASELINE.json configs[4], one GPU's rows of 100M x 128 on 8 GPUs"),
}


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
            summ = json.load(open(path))
            d = summ.get(kernel_name.replace(" ", "")) or summ.get(base)  # the exact instantiation when the summary has it
            if d and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                kib = 2.0 * d["FETCH_SIZE"]["avg_per_launch"] + d["WRITE_SIZE"]["avg_per_launch"]
                return kib * 1024.0, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def strided_init(n_global, m, k):
    import numpy as np

    return np.array([[(j * (n_global // k) + s) % n_global for j in range(k)] for s in range(m)], np.int64)


def mfma_roofline(flop, ms, kernel=None, extra=None):
    ach = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    r = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": ach / PEAK_F32_MFMA_TFLOPS, "avg_launch_ms": ms}
    if kernel:
        r["kernel"] = kernel
    if extra:
        r.update(extra)
    return r


def hbm_roofline(nbytes, ms, extra=None):
    ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    r = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
         "avg_launch_ms": ms, "algorithmic_bytes": nbytes}
    if extra:
        r.update(extra)
    return r


def other_configs(_lib, torch, engine):
    """BASELINE configs C1, C3, C4 on one GPU (rank 0, N=1): device-resident synthetic rows, codebooks from a few
    untimed Lloyd iterations, every entry with the roofline SURVEY.md 8(d) assigns it (HIP-event time of the
    assignment kernels from the library's profiling hooks; stream-synchronised wall time for whole iterations / builds)."""
    import numpy as np

    from vq_amd import TSVQ, Distance
    from vq_amd.tsvq import build_tree

    out = {}

    def pq_case(name, n, d, m, k, metric, label):
        ds = _lib.Dataset.synthetic(n, d, DATA_SEED, 0)
        km = _lib.KMeans(ds, m, k)
        km.set_engine(engine)
        km.init_from_rows(strided_init(n, m, k).astype(np.uint64))
        km.run(TRAIN_ITERS)
        km.run(10)
        _lib.synchronize()
        iters = 10
        t0 = time.perf_counter()
        it, _, _, paused = km.run(iters)
        _lib.synchronize()
        km_ms = (time.perf_counter() - t0) * 1e3 / max(1, int(it.max()))
        cb = km.get_centroids()
        km.close()
        enc = _lib.PQEncoder(cb, metric)
        enc.set_engine(engine)
        codes = torch.empty((n, m), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
        _lib.synchronize()
        _lib.set_profiling(True)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.encode_device(ds.device_ptr, n, codes.data_ptr(), None)
        _lib.synchronize()
        step_ms = (time.perf_counter() - t0) * 1e3 / reps
        calls, primary_ms, recheck_ms = _lib.profile_collect()
        _lib.set_profiling(False)
        rechecked, used = _lib.last_assign_stats()
        flop = 2.0 * k * d * n
        out[name] = {
            "workload": label, "rows": n, "dim": d, "m": m, "k": k, "sub_dim": d // m,
            "encode_vectors_per_s": n / (step_ms * 1e-3), "encode_ms_per_step": step_ms,
            "encode_engine": {1: "exact", 2: "fp32_mfma_screen", 3: "bf16x3_mfma_screen"}.get(used, str(used)),
            "recheck_fraction": rechecked / float(n * m),
            "encode_roofline": mfma_roofline(flop, primary_ms / max(calls, 1), extra={
                "step_frac": flop / (step_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "recheck_avg_launch_ms": recheck_ms / max(calls, 1), "flop_per_launch": flop}),
            "kmeans_ms_per_iter": km_ms, "kmeans_iter_per_s": 1e3 / km_ms,
            "kmeans_roofline": mfma_roofline(flop, km_ms, extra={
                "flop_per_iter": flop, "note": "whole Lloyd iteration (assign + fused update + reduce + finalize), decisions on the device"}),
        }
        enc.close()
        ds.close()

    pq_case("C1", 10_000, 64, 4, 16, _lib.EUCLIDEAN, "BASELINE.json configs[0]: PQ m=4 k=16 Euclidean, 10k x 64 (launch-latency bound on a GPU)")
    pq_case("C3", 1_000_000, 768, 96, 256, _lib.COSINE, "BASELINE.json configs[2]: PQ m=96 k=256 cosine, 1M x 768 (training is squared L2, src/core/vector.rs:352-363)")

    # C4: TSVQ depth 8 on 1M x 128
    n, d, depth = 1_000_000, 128, 8
    ds = _lib.Dataset.synthetic(n, d, DATA_SEED, 0)
    build_tree(ds, depth)
    ts = []
    for _ in range(5):
        _lib.synchronize()
        t0 = time.perf_counter()
        cent, left, right = build_tree(ds, depth)
        _lib.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    build_ms = sorted(ts)[len(ts) // 2]
    split_levels = depth
    build_bytes = 4.0 * n * d * (2 * split_levels + 1)  # mean + variance pass per split level, one leaf-mean pass
    t = TSVQ.from_tree(cent, left, right, Distance.euclidean())
    leaf = torch.empty(n, dtype=torch.int32, device="cuda")
    f16 = torch.empty((n, d), dtype=torch.float16, device="cuda")
    import ctypes as C

    lib = _lib.load()

    def enc_once():
        _lib.check(lib.vqhip_tsvq_encode_device(t._enc.raw, C.c_void_p(ds.device_ptr), n, C.c_void_p(leaf.data_ptr()),
                                                C.c_void_p(f16.data_ptr())))

    for _ in range(3):
        enc_once()
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        enc_once()
    _lib.synchronize()
    enc_ms = (time.perf_counter() - t0) * 1e3 / 10
    out["C4"] = {
        "workload": "BASELINE.json configs[3]: TSVQ depth 8 on 1M x 128, tree bit-identical to the reference's recursion",
        "rows": n, "dim": d, "depth": depth, "nodes": int(len(left)),
        "build_ms": build_ms, "build_ms_all": ts,
        "build_roofline": hbm_roofline(build_bytes, build_ms, {"note": "4*N*D bytes x (2 passes x 8 split levels + 1 leaf-mean pass)"}),
        "encode_vectors_per_s": n / (enc_ms * 1e-3), "encode_ms_per_step": enc_ms,
        "encode_roofline": hbm_roofline((4.0 * d + 2.0 * d) * n, enc_ms, {"note": "4*D bytes in + 2*D bytes (f16 reconstruction) out per vector"}),
    }
    ds.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="C2", help="workload (default: the metric's, C2)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the workload's)")
    ap.add_argument("--kmeans-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C3 / C4 block")
    ap.add_argument("--collective", choices=["native", "torch"], default="native",
                    help="multi-GPU all-reduce below the C ABI (vqhip_kmeans_run_sharded) or through torch.distributed")
    ap.add_argument("--engine", choices=["auto", "exact", "mfma", "bf16"], default="auto")
    args = ap.parse_args()
    wl = WORKLOADS[args.config]
    n, dim, m_, k_ = (args.rows or wl["rows"]), wl["dim"], wl["m"], wl["k"]
    # stdout carries exactly ONE line, the JSON: native libraries that write to fd 1 (RCCL prints a version banner when a
    # communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from vq_amd import _lib
    from vq_amd.sharded import Comm, HipShard, ShardedKMeans, native_comm_from_torch

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

    collective = "none"
    with torch.cuda.stream(stream):
        _lib.set_stream(stream.cuda_stream)
        n_global = n * world
        ds = _lib.Dataset.synthetic(n, dim, DATA_SEED, rank * n)
        init = strided_init(n_global, m_, k_)

        # ---- codebooks: a few (untimed) global Lloyd iterations from strided init rows; then the timed iterations ----
        use_native = args.collective == "native"
        ncomm = None
        if use_native:
            # the library's own RCCL communicator; if any rank cannot create it (RCCL missing, init error) every rank
            # falls back to torch.distributed's all-reduce of the same slab -- same bits, one more launch per iteration
            ok = 1
            try:
                ncomm = native_comm_from_torch(force=force_comm)  # identity communicator for one rank
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: native communicator unavailable ({e}); falling back to torch.distributed", file=sys.stderr)
                ok = 0
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                use_native = False
                if ncomm is not None:
                    ncomm.close()
        if use_native:
            collective = "rccl below the C ABI (vqhip_kmeans_run_sharded)" if (world > 1 or force_comm) else "none"
            km = _lib.KMeans(ds, m_, k_)
            km.set_engine(engine)
            km.init_from_global_rows(ncomm, init, rank * n)
            km.run(TRAIN_ITERS, ncomm)
            codebooks = km.get_centroids()
            for _ in range(3):  # untimed: the first ~30 iterations of a process run 10 % slow (clocks ramping up from idle)
                km.run(10, ncomm)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            it, _, _, paused = km.run(args.kmeans_iters, ncomm)
            torch.cuda.synchronize()
            barrier()
            km_dt = time.perf_counter() - t0
            km_iters = max(1, int(it.max()))
            km.close()
            ncomm.close()
        else:
            comm = Comm(force=force_comm)
            collective = "torch.distributed nccl" if comm.on else "none"
            shard = HipShard(ds, m_, k_, rank * n, engine)
            skm = ShardedKMeans(shard, n_global, comm)
            skm.init_from_global_rows(init)
            for _ in range(TRAIN_ITERS):
                skm.step()
            codebooks = shard.get_centroids()
            for _ in range(30):
                skm.step()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.kmeans_iters):
                skm.step()
            torch.cuda.synchronize()
            barrier()
            km_dt = time.perf_counter() - t0
            km_iters = args.kmeans_iters
            shard.close()
            _lib.set_stream(stream.cuda_stream)
        km_t = torch.tensor([km_dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(km_t, op=dist.ReduceOp.MAX)
        km_dt = float(km_t.item())

        # ---- encode: the timed region -------------------------------------------------------
        enc = _lib.PQEncoder(codebooks, _lib.SQUARED_EUCLIDEAN)
        enc.set_engine(engine)
        codes = torch.empty((n, m_), dtype=torch.uint8, device="cuda")
        f16 = torch.empty((n, dim), dtype=torch.float16, device="cuda")
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
        del f16

        extras = {}
        if rank == 0 and world == 1 and args.config == "C2":
            # (a) H2D-inclusive: host rows in, codes back to the host (never the headline)
            nh = min(n, 1_000_000)
            Xh = _lib.synth_uniform_host(nh, dim, DATA_SEED, 0)
            enc.encode(Xh, want_codes=True, want_f16=False)
            t0 = time.perf_counter()
            for _ in range(3):
                enc.encode(Xh, want_codes=True, want_f16=False)
            extras["encode_host_in_host_out_vectors_per_s"] = nh * 3 / (time.perf_counter() - t0)
            extras["encode_host_rows"] = nh
            del Xh
            # (b) clustered data once (mixture of K Gaussians around the trained centroids' scale):
            # uniform data is the worst case for near-ties, this is the friendly one
            g = torch.Generator(device="cuda").manual_seed(DATA_SEED)
            centers = torch.rand((k_, dim), device="cuda", generator=g)
            which = torch.randint(0, k_, (n,), device="cuda", generator=g)
            Xc = (centers[which] + 0.02 * torch.randn((n, dim), device="cuda", generator=g)).contiguous()
            torch.cuda.synchronize()
            dsc = _lib.Dataset.from_device(Xc.data_ptr(), n, dim)
            kmc = _lib.KMeans(dsc, m_, k_)
            kmc.set_engine(engine)
            kmc.init_from_rows(strided_init(n, m_, k_).astype(np.uint64))
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
            extras["clustered_data_recheck_fraction"] = _lib.last_assign_stats()[0] / float(n * m_)
            encc.close()
            kmc.close()
            dsc.close()
            del Xc
        enc.close()
        ds.close()
        del codes
        configs = None
        if rank == 0 and world == 1 and not args.no_configs:
            configs = other_configs(_lib, torch, engine)

    if rank == 0:
        value = n_global * args.steps / dt
        flop_per_row = 2.0 * k_ * dim  # the -2.x.c contraction only
        kern_s = primary_ms / 1e3 / max(calls, 1)
        achieved = flop_per_row * n / kern_s / 1e12 if kern_s > 0 else 0.0
        sd = dim // m_
        kernel_name = {2: f"k_assign_screen<{sd},16>", 3: f"k_assign_screen_bf16_x32<{sd},8,1,0,false>"}.get(used_engine, "k_assign_exact")
        traffic, traffic_src = pmc_traffic(kernel_name) if (n == N_PER_GPU and args.config == "C2") else (None, None)
        step_ms = dt / args.steps * 1e3
        km_ms = km_dt / km_iters * 1e3
        line = {
            "metric": "pq_encode_vectors_per_s",
            "value": value,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"PQ m={m_} k={k_} L2 encode on {n}x{dim} f32 rows per GPU, device-resident "
                            f"({wl['label']}); codes out (1 byte per subspace)",
                "rows_per_gpu": n, "dim": dim, "m": m_, "k": k_, "sub_dim": sd,
                "engine": {1: "exact", 2: "fp32_mfma_screen+exact_recheck",
                           3: "bf16x3_mfma_screen+exact_recheck"}.get(used_engine, str(used_engine)),
                "recheck_fraction": rechecked / float(n * m_),
                "codebooks": f"{TRAIN_ITERS} Lloyd iterations from strided init rows",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": kernel_name,
                "note": "achieved = algorithmic 2*k*D flop per row / device time of the screen kernel (HIP events on the "
                        "launch stream); with the bf16-split engine the contraction runs as 6 bf16 products per fp32 "
                        "product on the bf16 matrix pipe and the kernel is VALU-bound (epilogue), see DESIGN.md 4.1; "
                        "step_frac = the same work over the whole driver-timed step (screen + exact re-check)",
                "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                "step_frac": flop_per_row * n / (step_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "flop_per_launch": flop_per_row * n,
                "avg_launch_ms": kern_s * 1e3,
                "recheck_avg_launch_ms": recheck_ms / max(calls, 1),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": (4.0 * dim + m_) * n,
            },
            "kmeans_iter_per_s": km_iters / km_dt,
            "kmeans_ms_per_iter": km_ms,
            "kmeans_rows_global": n_global,
            "kmeans_collective": collective,
            "kmeans_roofline": mfma_roofline(flop_per_row * n_global / world, km_ms, extra={
                "flop_per_iter_per_gpu": flop_per_row * n,
                "note": "per GPU: 2*N*k*D flop of one Lloyd iteration (assign + fused update + all-reduce + finalize) / its wall "
                        "time; X is read once per iteration (4*N*D bytes, SURVEY.md 8(d))"}),
            "encode_f16_out_vectors_per_s_per_gpu": n / dt_f16,
            "codes_checksum_rank0": checksum,
        }
        line.update(extras)
        if configs is not None:
            line["configs"] = configs
        r32, r16, trusted = _lib.selftest()
        line["bf16_mfma_selftest"] = {"ratio_32x32x16": r32, "ratio_16x16x32": r16, "model_bound": 18.1, "budget": 20.0,
                                      "trusted": trusted, "basis": "hardware == bit-exact adder model on 2^22 operand sets"}
        if world == 1 and not args.no_cpu_baseline and args.config == "C2":
            line["cpu_baseline"] = cpu_baseline(m_, k_, dim, codebooks)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or force_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
