"""Row-sharded k-means across the GPUs of one node: one process per GPU, rows resident on
their GPU for the whole fit, ONE collective per Lloyd iteration.

The reference has no multi-device path (its only parallelism is rayon over rows in the
assignment step, src/core/vector.rs:417-423).  The shard-level equivalent: every rank runs
assign + accumulate on its own rows, the per-cluster partial sums and counts -- one fused
f64 slab [m][k][sub_dim+1], 139 KB at m=8, k=256, sub_dim=16 -- are summed with a single
all-reduce (RCCL over xGMI when the process group is ``nccl``), and every rank then computes
the same means and the same ``changed`` flags, so no second collective is needed.

Empty-cluster reseeds (vector.rs:448-452) name a GLOBAL row id; the rank that owns the row
broadcasts its sub-vector and all ranks patch the same centroid.

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
    def local_sub_row(self, s: int, global_row: int) -> np.ndarray: ...
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
    """torch.distributed wrapper that degrades to a no-op for a single process."""

    def __init__(self, force: bool = False):
        """force=True keeps the collectives on for a 1-rank group (used to exercise RCCL on one GPU)"""
        import torch.distributed as dist

        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1

    def all_reduce_sum(self, t):
        if self.on:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def broadcast(self, t, src: int):
        if self.on:
            self.dist.broadcast(t, src=src)

    def barrier(self):
        if self.on:
            self.dist.barrier()


class ShardedKMeans:
    def __init__(self, backend: ShardBackend, n_global: int, comm: Comm | None = None):
        self.b = backend
        self.n_global = int(n_global)
        self.comm = comm or Comm()
        self.active = np.ones(backend.m, dtype=bool)
        self.iters = np.zeros(backend.m, dtype=np.int64)

    # -- helpers ------------------------------------------------------------------------
    def _gather_global_rows(self, rows: np.ndarray) -> np.ndarray:
        """centroid values [m][k][sd] for global row ids [m][k]: each rank fills what it owns,
        one all-reduce(sum) assembles the rest (every entry has exactly one owner)."""
        import torch

        b = self.b
        vals = np.zeros((b.m, b.k, b.sd), np.float64)
        for s in range(b.m):
            for j in range(b.k):
                r = int(rows[s, j])
                if b.row_offset <= r < b.row_offset + b.n_local:
                    vals[s, j] = b.local_sub_row(s, r)
        t = torch.from_numpy(vals)
        dev = self._collective_device()
        if dev is not None:
            t = t.to(dev)
        self.comm.all_reduce_sum(t)
        return t.cpu().numpy().astype(np.float32)

    def _collective_device(self):
        if not self.comm.on:
            return None
        backend = self.comm.dist.get_backend()
        if backend == "nccl":
            import torch

            return torch.device("cuda", torch.cuda.current_device())
        return None

    def _bcast_sub_row(self, s: int, global_row: int) -> np.ndarray:
        import torch

        b = self.b
        owner = owner_of(self.n_global, self.comm.world, global_row)
        if self.comm.rank == owner:
            v = torch.from_numpy(np.ascontiguousarray(b.local_sub_row(s, global_row), np.float32))
        else:
            v = torch.zeros(b.sd, dtype=torch.float32)
        dev = self._collective_device()
        if dev is not None:
            v = v.to(dev)
        self.comm.broadcast(v, owner)
        return v.cpu().numpy()

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
    """libvqhip back end of one rank: a resident Dataset shard + a vqhip_kmeans handle."""

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

    def init_from_values(self, centroids):
        self.km.set_centroids(centroids)

    def local_sub_row(self, s, global_row):
        r = self.ds.read(global_row - self.row_offset, 1)[0]
        return r[s * self.sd:(s + 1) * self.sd]

    def accumulate(self):
        self.km.accumulate()

    def slab(self):
        import torch

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
        if not self._slab_alias:
            ptr, n = self.km.partials()
            self._lib.memcpy_device(ptr, self._slab_t.data_ptr(), n * 8)

    def finalize(self):
        return self.km.finalize()

    def patch_centroid(self, s, j, sub_row):
        self.km.patch_centroid(s, j, sub_row)

    def set_active(self, active):
        self.km.set_active(active)

    def get_centroids(self):
        return self.km.get_centroids()

    def close(self):
        self.km.close()
