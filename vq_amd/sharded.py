"""Row-sharded k-means across the GPUs of one node: one process per GPU, rows resident on
their GPU for the whole fit, ONE collective per Lloyd iteration.

The reference has no multi-device path (its only parallelism is rayon over rows in the
assignment step, src/core/vector.rs:417-423).  The shard-level equivalent: every rank runs
assign + accumulate on its own rows, the per-cluster partial sums and counts -- one fused
f64 slab [m][k][sub_dim+1], 139 KB at m=8, k=256, sub_dim=16 -- are summed with a single
all-reduce (RCCL over xGMI when the process group is ``nccl``), and every rank then computes
the same means and the same ``changed`` flags, so no second collective is needed.

Initial centroids and empty-cluster reseeds (vector.rs:412-413, 448-452) name GLOBAL row ids;
the rank that owns a row contributes its bit pattern, everyone else zero words, and an integer
sum all-reduce hands the bits to all ranks (ONE device gather and ONE copy per rank; a float sum
would turn -0.0 into +0.0).

Two ways to run the collective, same results:
  * ``ShardedKMeans(HipShard(...), n, Comm())``: torch.distributed (``nccl`` = RCCL, or ``gloo``
    on CPU ranks in tests).  ``HipShard`` puts the library on torch's current stream for every
    call, so the all-reduce is ordered behind ``accumulate`` and in front of ``finalize``.
  * ``NativeShardedKMeans(ds, m, k, n, native_comm)``: the collective runs BELOW the C ABI
    (``vqhip_kmeans_step_sharded``, include/vqhip.h) on the library's own stream -- what a Rust or
    C host calls; torch.distributed is used only to hand the RCCL unique id to the ranks.

``ShardedKMeans`` only orchestrates; the device work sits behind a small back-end protocol
(``HipShard`` = libvqhip).  tests/ drive the same orchestration on CPU ranks (gloo) with a
test-only back end.
"""
from __future__ import annotations

from typing import Protocol, Sequence

import numpy as np

from .rng import HostRng


class ShardBackend(Protocol):
    m: int
    k: int
    sd: int
    n_local: int
    row_offset: int

    def init_from_values(self, centroids: np.ndarray) -> None: ...
    def owned_bits(self, rows: np.ndarray) -> np.ndarray: ...  # uint32 [m][k][sd]: bits of owned rows, 0 elsewhere
    def owned_sub_row_bits(self, s: int, global_row: int) -> np.ndarray: ...  # uint32 [sd]
    def accumulate(self) -> None: ...
    def slab(self): ...  # torch tensor aliasing (or staging) the f64 slab
    def commit_slab(self) -> None: ...  # write a staged slab back (no-op when aliased)
    def finalize(self) -> tuple[np.ndarray, np.ndarray]: ...
    def patch_centroid(self, s: int, j: int, sub_row: np.ndarray) -> None: ...
    def set_active(self, active: np.ndarray) -> None: ...
    def get_centroids(self) -> np.ndarray: ...


def shard_rows(n_global: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous row block of `rank`: (offset, count); the first n%world ranks get one more"""
    base, rem = divmod(n_global, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def owner_of(n_global: int, world: int, row: int) -> int:
    base, rem = divmod(n_global, world)
    cut = rem * (base + 1)
    if row < cut:
        return row // (base + 1)
    return rem + (row - cut) // base


class Comm:
    """torch.distributed wrapper that degrades to a no-op for a single process.  `group`: the process group the
    collectives run on (default: the default group)."""

    def __init__(self, force: bool = False, group=None):
        """force=True keeps the collectives on for a 1-rank group (used to exercise RCCL on one GPU)"""
        import torch.distributed as dist

        self.dist, self.group = dist, group
        up = dist.is_available() and dist.is_initialized()
        self.on = up and (dist.get_world_size(group) > 1 or force)
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.backend = dist.get_backend(group) if self.on else None

    def all_reduce_sum(self, t):
        if not self.on:
            return
        if self.backend == "gloo" and t.is_cuda:  # host-staged (CPU ranks' transport; tests and one-GPU boxes)
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def broadcast(self, t, src: int):
        if self.on:
            self.dist.broadcast(t, src=src, group=self.group)

    def barrier(self):
        if self.on:
            self.dist.barrier(group=self.group)


class ShardedKMeans:
    def __init__(self, backend: ShardBackend, n_global: int, comm: Comm | None = None):
        self.b = backend
        self.n_global = int(n_global)
        self.comm = comm or Comm()
        self.active = np.ones(backend.m, dtype=bool)
        self.iters = np.zeros(backend.m, dtype=np.int64)

    # -- helpers ------------------------------------------------------------------------
    def _sum_bits(self, bits: np.ndarray) -> np.ndarray:
        """integer all-reduce(sum) of uint32 words with exactly one non-zero contributor each"""
        import torch

        t = torch.from_numpy(np.ascontiguousarray(bits, np.uint32).view(np.int32).copy())
        dev = self._collective_device()
        if dev is not None:
            t = t.to(dev)
        self.comm.all_reduce_sum(t)
        return t.cpu().numpy().view(np.uint32)

    def _gather_global_rows(self, rows: np.ndarray) -> np.ndarray:
        """centroid values [m][k][sd] for global row ids [m][k]: each rank supplies the bits of
        the rows it owns (one gather on its device), one all-reduce assembles the rest"""
        b = self.b
        return self._sum_bits(b.owned_bits(rows)).view(np.float32).reshape(b.m, b.k, b.sd)

    def _collective_device(self):
        if not self.comm.on:
            return None
        if self.comm.backend == "nccl":
            import torch

            return torch.device("cuda", torch.cuda.current_device())
        return None

    def _bcast_sub_row(self, s: int, global_row: int) -> np.ndarray:
        return self._sum_bits(self.b.owned_sub_row_bits(s, global_row)).view(np.float32)

    # -- API ----------------------------------------------------------------------------
    def init_from_global_rows(self, rows) -> None:
        rows = np.asarray(rows, dtype=np.int64).reshape(self.b.m, self.b.k)
        self.b.init_from_values(self._gather_global_rows(rows))

    def step(self) -> tuple[np.ndarray, np.ndarray]:
        """one Lloyd iteration over the global data set; returns (counts [m][k], changed [m])"""
        self.b.accumulate()
        if self.comm.on:
            t = self.b.slab()
            self.comm.all_reduce_sum(t)
            self.b.commit_slab()
        counts, changed = self.b.finalize()
        return counts, changed

    def fit(self, max_iters: int, seed: int = 42, init_rows=None,
            reseed_rows: Sequence[Sequence[int]] | None = None) -> np.ndarray:
        """control flow of lbg_quantize (src/core/vector.rs:412-460) for all subspaces; every
        rank executes it identically (same RNG seeds, same all-reduced counts/changed)."""
        b = self.b
        n = self.n_global
        rngs = [HostRng(seed + s) for s in range(b.m)]
        if init_rows is None:
            init_rows = np.array([rngs[s].choose_multiple(n, b.k) for s in range(b.m)], np.int64)
        self.init_from_global_rows(init_rows)
        reseed_it = None if reseed_rows is None else [iter(list(r)) for r in reseed_rows]
        self.active[:] = True
        self.iters[:] = 0
        for _ in range(max_iters):
            if not self.active.any():
                break
            counts, changed = self.step()
            self.iters[self.active] += 1
            for s, j in np.argwhere((counts == 0) & self.active[:, None]):  # (subspace, ascending j)
                row = int(next(reseed_it[s])) if reseed_it is not None else rngs[s].choose(n)
                b.patch_centroid(int(s), int(j), self._bcast_sub_row(int(s), row))
            converged = self.active & ~np.asarray(changed, dtype=bool)
            if converged.any():
                self.active[converged] = False
                b.set_active(self.active)
        return b.get_centroids()


class HipShard:
    """libvqhip back end of one rank: a resident Dataset shard + a vqhip_kmeans handle.

    Stream ordering: the torch collective runs on (or is ordered against) torch's CURRENT stream,
    the library enqueues on the calling thread's vqhip stream.  Every method therefore first puts
    the library on torch's current stream (``vqhip_set_stream``), which orders
    accumulate -> all_reduce -> finalize without any host synchronisation."""

    def __init__(self, ds, m: int, k: int, row_offset: int, engine: int = 0):
        from . import _lib

        self._lib = _lib
        self.ds = ds
        self.km = _lib.KMeans(ds, m, k)
        self.km.set_engine(engine)
        self.m, self.k, self.sd = m, k, ds.d // m
        self.n_local, self.row_offset = ds.n, int(row_offset)
        self._slab_t = None
        self._slab_alias = False

    def _on_torch_stream(self):
        import torch

        self._lib.set_stream(torch.cuda.current_stream().cuda_stream)

    def init_from_values(self, centroids):
        self._on_torch_stream()
        self.km.set_centroids(centroids)

    def owned_bits(self, rows):
        self._on_torch_stream()
        return self.km.gather_owned_rows(rows, self.row_offset)

    def owned_sub_row_bits(self, s, global_row):
        r = int(global_row) - self.row_offset
        if not 0 <= r < self.n_local:
            return np.zeros(self.sd, np.uint32)
        self._on_torch_stream()
        return self.ds.read(r, 1)[0, s * self.sd:(s + 1) * self.sd].view(np.uint32).copy()

    def accumulate(self):
        self._on_torch_stream()
        self.km.accumulate()

    def slab(self):
        import torch

        self._on_torch_stream()
        ptr, n = self.km.partials()
        if self._slab_t is None:
            try:  # zero-copy view of the library's device buffer
                class _Iface:
                    pass

                holder = _Iface()
                holder.__cuda_array_interface__ = {
                    "shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
                t = torch.as_tensor(holder, device=torch.device("cuda", torch.cuda.current_device()))
                if t.data_ptr() != ptr:
                    raise RuntimeError("not aliased")
                self._slab_t, self._slab_alias = t, True
            except Exception:
                self._slab_t = torch.empty(n, dtype=torch.float64, device="cuda")
                self._slab_alias = False
        if not self._slab_alias:
            self._lib.memcpy_device(self._slab_t.data_ptr(), ptr, n * 8)
        return self._slab_t

    def commit_slab(self):
        self._on_torch_stream()
        if not self._slab_alias:
            ptr, n = self.km.partials()
            self._lib.memcpy_device(ptr, self._slab_t.data_ptr(), n * 8)

    def finalize(self):
        self._on_torch_stream()
        return self.km.finalize()

    def patch_centroid(self, s, j, sub_row):
        self._on_torch_stream()
        self.km.patch_centroid(s, j, sub_row)

    def set_active(self, active):
        self._on_torch_stream()
        self.km.set_active(active)

    def get_centroids(self):
        self._on_torch_stream()
        return self.km.get_centroids()

    def close(self):
        self.km.close()


def native_comm_from_torch(force: bool = False):
    """vqhip_comm (RCCL below the C ABI) for the ranks of the initialised torch.distributed group:
    rank 0 draws the RCCL unique id, torch's store-backed object broadcast hands it round, every
    rank then joins with ncclCommInitRank on its current HIP device.  Without a process group (or
    with one rank and force=False) the result is the identity communicator."""
    from . import _lib

    try:
        import torch.distributed as dist

        on = dist.is_available() and dist.is_initialized()
    except Exception:  # pragma: no cover
        on = False
    if not on or (dist.get_world_size() == 1 and not force):
        return _lib.NativeComm(None, 1, 0)
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [_lib.NativeComm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return _lib.NativeComm(box[0], world, rank)


class NativeShardedKMeans:
    """Row-sharded Lloyd with the collective below the C ABI: each call is ONE library entry point
    (vqhip_kmeans_init_from_global_rows / _step_sharded / _patch_from_global_row), exactly the
    sequence a Rust host would issue.  Control flow of lbg_quantize
    (src/core/vector.rs:412-460); every rank takes identical decisions from the global counts."""

    def __init__(self, ds, m: int, k: int, n_global: int, row_offset: int, comm, engine: int = 0):
        from . import _lib

        self.km = _lib.KMeans(ds, m, k)
        self.km.set_engine(engine)
        self.m, self.k, self.sd = m, k, ds.d // m
        self.n_global, self.row_offset, self.comm = int(n_global), int(row_offset), comm
        self.active = np.ones(m, dtype=bool)
        self.iters = np.zeros(m, dtype=np.int64)

    def init_from_global_rows(self, rows) -> None:
        self.km.init_from_global_rows(self.comm, rows, self.row_offset)

    def step(self):
        return self.km.step_sharded(self.comm)

    def fit(self, max_iters: int, seed: int = 42, init_rows=None, reseed_rows=None) -> np.ndarray:
        n = self.n_global
        rngs = [HostRng(seed + s) for s in range(self.m)]
        if init_rows is None:
            init_rows = np.array([rngs[s].choose_multiple(n, self.k) for s in range(self.m)], np.int64)
        self.init_from_global_rows(init_rows)
        reseed_it = None if reseed_rows is None else [iter(list(r)) for r in reseed_rows]
        self.active[:] = True
        self.km.set_active(self.active)
        self.iters[:] = 0
        done = 0
        while done < max_iters and self.active.any():
            it, counts, changed, paused = self.km.run(max_iters - done, self.comm)  # vqhip_kmeans_run_sharded
            self.iters += it.astype(np.int64)
            done += max(1, int(it.max()))
            # the library's set: converged subspaces retire on the device, also iterations BEFORE a pause (their
            # `counts` then read 0 and must not be taken for empty clusters)
            self.active = self.km.get_active()
            if not paused:
                continue
            for s, j in np.argwhere((counts == 0) & self.active[:, None]):
                row = int(next(reseed_it[s])) if reseed_it is not None else rngs[s].choose(n)
                self.km.patch_from_global_row(self.comm, int(s), int(j), row, self.row_offset)
            converged = self.active & ~np.asarray(changed, dtype=bool)
            if converged.any():
                self.active[converged] = False
                self.km.set_active(self.active)
        return self.km.get_centroids()

    def close(self):
        self.km.close()
