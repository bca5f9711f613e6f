"""``ProductQuantizer`` -- host mirror of the reference's public PQ interface.

Same constructor keywords, defaults, properties, ``repr`` and error text as
pyvq.ProductQuantizer (reference pyvq/src/pq.rs:48-155) / ``ProductQuantizer::new``
(src/pq.rs:83-141); the bodies run on the MI355X through libvqhip:

* ``__init__``      -> vqhip_kmeans_* : Lloyd iterations for all m subspaces at once (the
                       reference runs them one after the other, src/pq.rs:121); each subspace
                       still converges on its own and draws its own init/reseed rows from an
                       RNG seeded with ``seed + s`` (src/pq.rs:130).
* ``quantize``      -> vqhip_pq_encode : first-minimum argmin per subspace, selected centroid
                       as float16 (src/pq.rs:167-199).
* new, batch-shaped: ``quantize_batch``, ``encode`` (the internal ``best_idx`` codes),
                       ``decode``.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .distance import Distance
from .errors import DimensionMismatch, EmptyInput, InvalidParameter
from .rng import HostRng


def _as_training_matrix(training_data) -> np.ndarray:
    """``&[&[f32]]`` -> one contiguous [n][d] f32 matrix with the reference's checks
    (src/pq.rs:91-104, pyvq/src/pq.rs:59-62)."""
    if isinstance(training_data, np.ndarray):
        if training_data.ndim != 2:
            raise ValueError("training_data must be a 2D array")
        if training_data.shape[0] == 0:
            raise ValueError("Training data cannot be empty")  # pyvq/src/pq.rs:61
        return np.ascontiguousarray(training_data, dtype=np.float32)
    rows = list(training_data)
    if len(rows) == 0:
        raise EmptyInput()
    dim = len(rows[0])
    for r in rows:
        if len(r) != dim:
            raise DimensionMismatch(dim, len(r))
    return np.ascontiguousarray(np.asarray(rows, dtype=np.float32).reshape(len(rows), dim))


def fit_codebooks(ds: "_lib.Dataset | _lib.MDataset", m: int, k: int, max_iters: int, seed: int = 42, *,
                  init_rows=None, reseed_rows=None, engine: int = _lib.ENGINE_AUTO,
                  exact_update: bool = False, stats: dict | None = None) -> np.ndarray:
    """Lloyd / LBG for all m subspaces of a resident dataset: the control flow of
    ``lbg_quantize`` (src/core/vector.rs:396-460) with the device doing each iteration.

    ``init_rows`` [m][k] and ``reseed_rows`` (m sequences consumed in order) replace the
    package's own RNG draws: identical draws give the reference's assignment codes bit for bit
    and its centroids within ``1e-5 * max(1, |c|)`` per step -- bit for bit, iteration counts
    included, with ``exact_update=True`` (sums in the reference's row order; single GPU).
    """
    n, d = ds.n, ds.d
    # validation order and messages: src/pq.rs:106-117 then src/core/vector.rs:396-410
    if m == 0:
        raise InvalidParameter("m", "must be greater than 0")
    if d < m:
        raise InvalidParameter("m", f"must be at most the data dimension ({d})")
    if d % m != 0:
        raise InvalidParameter("m", f"dimension ({d}) must be divisible by m")
    if k == 0:
        raise InvalidParameter("k", "must be greater than 0")
    if n < k:
        raise InvalidParameter("k", f"not enough data points ({n}) for {k} clusters")
    rngs = [HostRng(seed + s) for s in range(m)]  # seed + i, src/pq.rs:130
    if init_rows is None:
        init = np.array([rngs[s].choose_multiple(n, k) for s in range(m)], dtype=np.uint64)
    else:
        init = np.ascontiguousarray(init_rows, dtype=np.uint64).reshape(m, k)
    reseed_iters = None if reseed_rows is None else [iter(list(r)) for r in reseed_rows]

    # rows resident on one device, or in blocks over several devices of this process (the ranks of the sharded fit run as
    # worker threads inside the library: same methods, row ids global)
    km = _lib.MKMeans(ds, m, k) if isinstance(ds, _lib.MDataset) else _lib.KMeans(ds, m, k)
    try:
        km.set_engine(engine)
        km.set_exact_update(exact_update)
        km.init_from_rows(init)
        active = np.ones(m, dtype=bool)
        iters = np.zeros(m, dtype=np.int64)
        n_reseeds = 0
        done = 0
        while done < max_iters and active.any():  # vector.rs:415
            # the loop's decisions are taken on the device (vqhip_kmeans_run): it comes back when the iterations are
            # used up, when every subspace has converged, or PAUSED after an iteration that left a cluster empty
            it, counts, changed, paused = km.run(max_iters - done)
            iters += it.astype(np.int64)
            done += max(1, int(it.max()))
            # the library's own set: subspaces that converged inside this call were retired on the device
            # (vector.rs:455-457) -- also those that converged iterations BEFORE a pause, whose `counts` read 0
            active = km.get_active()
            if not paused:
                continue
            # empty clusters of the subspaces that executed the pausing iteration, in (subspace, ascending j) order,
            # vector.rs:448-452 (one vectorised scan: the per-subspace Python loop cost ~1 ms per iteration at m = 96)
            empties = np.argwhere((counts == 0) & active[:, None])
            for s, j in empties:
                if reseed_iters is not None:
                    try:
                        row = int(next(reseed_iters[s]))
                    except StopIteration:
                        raise InvalidParameter("reseed_rows", f"exhausted for subspace {s}")
                else:
                    row = rngs[s].choose(n)
                km.patch_from_row(int(s), int(j), row)
                n_reseeds += 1
            converged = active & ~np.asarray(changed, dtype=bool)  # vector.rs:455-457
            if converged.any():
                active[converged] = False
                km.set_active(active)
        if stats is not None:
            stats["iters"] = iters
            stats["reseeds"] = n_reseeds
        return km.get_centroids()
    finally:
        km.close()


class ProductQuantizer:
    """Product quantizer: m subspaces, k centroids each.

    Args (pyvq/src/pq.rs:36-49): training_data (n, dim) float32; num_subspaces (m);
    num_centroids (k); max_iters=10; distance=None (-> Euclidean); seed=42.
    """

    def __init__(self, training_data, num_subspaces: int, num_centroids: int, max_iters: int = 10,
                 distance: Distance | None = None, seed: int = 42, *, init_rows=None,
                 reseed_rows=None, engine: int = _lib.ENGINE_AUTO, exact_update: bool = False, devices=None):
        """devices (not in pyvq): the GPUs the training batch is partitioned over -- None: every visible device the batch
        gives 4M elements of work to (ONE device under ``exact_update``: the reference's summation order is a single
        chain over all rows); an int n: devices 0..n-1; a sequence of device ids.  One call, one process: with more than
        one device the rows are sharded over worker threads inside the library and each Lloyd iteration all-reduces the
        per-cluster sums (include/vqhip.h, "one call, one process, several GPUs"); batch encodes, ``decode`` and
        ``dequantize_batch`` split their rows the same way.  Codes are the same bits on any device count GIVEN the
        codebooks; the trained centroids are not: the per-cluster sums are combined per device, then over devices (within
        the 1e-5 step tolerance of one device's, deterministic for a given device list) -- pass ``devices=[0]`` for
        codebooks that do not depend on the machine."""
        X = _as_training_matrix(training_data)
        n, dim = X.shape
        m, k = int(num_subspaces), int(num_centroids)
        if m == 0:
            raise InvalidParameter("m", "must be greater than 0")
        if dim < m:
            raise InvalidParameter("m", f"must be at most the data dimension ({dim})")
        if dim % m != 0:
            raise InvalidParameter("m", f"dimension ({dim}) must be divisible by m")
        if k == 0:
            raise InvalidParameter("k", "must be greater than 0")
        if n < k:
            raise InvalidParameter("k", f"not enough data points ({n}) for {k} clusters")
        self._distance = distance if distance is not None else Distance.euclidean()
        self._m, self._k, self._dim, self._sub_dim = m, k, dim, dim // m
        self.fit_stats: dict = {}
        dv = _lib._devices(devices)
        if devices is None:  # every visible device, as far as the batch gives each of them work (4M elements)
            dv = dv[:max(1, min(dv.size, (n * dim) >> 22))]
            if exact_update:  # one sequential chain over all rows: the one-device path (an explicit list still raises)
                dv = dv[:1]
        if dv.size > n:
            dv = dv[:n]  # (every device needs a row)
        # one slot naming the calling thread's current device: the single-device handles; anything else -- several slots, or
        # one slot on ANOTHER device -- goes through the one-process multi-device handles, whose worker threads own their
        # devices (one slot there is the single-device path bit for bit, tests/test_gpu_multi.py; ADVICE r5: `devices=[1]`
        # used to run on the current device and report 1)
        self._pinned = dv.size == 1 and int(dv[0]) != _lib.get_device()
        multi = dv.size > 1 or self._pinned
        ds = _lib.MDataset.from_host(X, dv) if multi else _lib.Dataset.from_host(X)
        try:
            self._codebooks = fit_codebooks(ds, m, k, int(max_iters), int(seed), init_rows=init_rows,
                                            reseed_rows=reseed_rows, engine=engine,
                                            exact_update=exact_update, stats=self.fit_stats)
        finally:
            ds.close()
        self.fit_stats["devices"] = [int(x) for x in dv]
        self._enc = _lib.PQEncoder(self._codebooks, self._distance.metric)  # (current device: per-vector calls, ADC search)
        self._enc.set_engine(engine)
        self._menc = None
        if multi:  # batch encodes: row blocks over the same devices
            self._menc = _lib.MPQEncoder(self._codebooks, self._distance.metric, dv)
            self._menc.set_engine(engine)

    @classmethod
    def from_codebooks(cls, codebooks, distance: Distance | None = None,
                       engine: int = _lib.ENGINE_AUTO) -> "ProductQuantizer":
        """Wrap existing codebooks [m][k][sub_dim] (no training)."""
        self = cls.__new__(cls)
        cb = np.ascontiguousarray(codebooks, dtype=np.float32)
        if cb.ndim != 3:
            raise ValueError("codebooks must have shape (m, k, sub_dim)")
        self._m, self._k, self._sub_dim = (int(x) for x in cb.shape)
        self._dim = self._m * self._sub_dim
        self._distance = distance if distance is not None else Distance.euclidean()
        self._codebooks = cb
        self.fit_stats = {}
        self._enc = _lib.PQEncoder(cb, self._distance.metric)
        self._enc.set_engine(engine)
        self._menc = None
        self._pinned = False
        return self

    def _batch_encoder(self, n: int):
        # (small batches are not worth the workers' hand-over -- unless the quantizer was pinned to another device)
        return self._menc if (self._menc is not None and (n >= 65536 or self._pinned)) else self._enc

    # -- reference surface ----------------------------------------------------------------
    def quantize(self, vector) -> np.ndarray:
        """float32 (dim,) -> float16 (dim,): the selected centroids (src/pq.rs:167-199)"""
        v = np.ascontiguousarray(vector, dtype=np.float32).ravel()
        if v.size != self._dim:
            raise DimensionMismatch(self._dim, v.size)
        _, f16 = self._batch_encoder(1).encode(v[None, :], want_codes=False, want_f16=True)
        return f16[0]

    def dequantize(self, codes) -> np.ndarray:
        """float16 (dim,) -> float32 (dim,) (src/pq.rs:201-209)"""
        q = np.ascontiguousarray(codes, dtype=np.float16).ravel()
        if q.size != self._dim:
            raise DimensionMismatch(self._dim, q.size)
        return _lib.dequantize_f16(q)

    @property
    def num_subspaces(self) -> int:
        return self._m

    @property
    def sub_dim(self) -> int:
        return self._sub_dim

    @property
    def dim(self) -> int:
        return self._dim

    def distance_metric(self) -> str:
        return self._distance.name()

    def __repr__(self) -> str:  # pyvq/src/pq.rs:147-154
        return (f"ProductQuantizer(dim={self._dim}, num_subspaces={self._m}, "
                f"sub_dim={self._sub_dim})")

    # -- batch additions (ROADMAP.md:30 "batch quantization" is open upstream) --------------
    @property
    def num_centroids(self) -> int:
        return self._k

    @property
    def codebooks(self) -> np.ndarray:
        return self._codebooks

    def _check_batch(self, X) -> np.ndarray:
        X = np.ascontiguousarray(X, dtype=np.float32)
        if X.ndim != 2:
            raise ValueError("expected a 2D array (n, dim)")
        if X.shape[1] != self._dim:
            raise DimensionMismatch(self._dim, X.shape[1])
        return X

    def quantize_batch(self, X, out=None) -> np.ndarray:
        """(n, dim) float32 -> (n, dim) float16, row i == quantize(X[i]).  out: a float16 array (n, dim) to fill (numpy's
        `out=` convention; a reused array skips the first-touch page faults of a fresh one, 25 of 35 ms at 1M x 128)"""
        X = self._check_batch(X)
        if X.shape[0] == 0:
            return np.empty((0, self._dim), np.float16) if out is None else out
        return self._batch_encoder(X.shape[0]).encode(X, want_codes=False, want_f16=True, out_f16=out)[1]

    def encode(self, X) -> np.ndarray:
        """(n, dim) float32 -> (n, m) codes: ``best_idx`` per subspace (src/pq.rs:183-191); uint8 while
        k <= 256, uint16 above"""
        X = self._check_batch(X)
        if X.shape[0] == 0:
            return np.empty((0, self._m), _lib.code_dtype(self._k))
        return self._batch_encoder(X.shape[0]).encode(X, want_codes=True, want_f16=False)[0]

    def search(self, codes, queries, topk: int = 10):
        """Asymmetric-distance search: the `topk` rows of `codes` (n, m) (uint8, uint16 above 256 centroids)
        nearest to each query under this quantizer's metric, distances from per-subspace tables.
        Returns (indices uint32 (nq, topk), distances float32 (nq, topk)); ties by lower row."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self._dim:
            raise DimensionMismatch(self._dim, q.shape[1])
        codes = np.ascontiguousarray(codes, dtype=_lib.code_dtype(self._k))
        if codes.ndim != 2 or codes.shape[1] != self._m:
            raise DimensionMismatch(self._m, codes.shape[1] if codes.ndim == 2 else codes.size)
        if not 1 <= topk <= min(codes.shape[0], 1024):
            raise InvalidParameter("topk", f"must be between 1 and min(n, 1024), got {topk}")
        if self._distance.metric in (_lib.COSINE, _lib.COSINE_UNCLAMPED):
            raise InvalidParameter("distance", "cosine distance is not a sum over subspaces: no ADC form")
        if codes.size and int(codes.max()) >= self._k:  # the scan indexes its LDS tables by code
            raise InvalidParameter("codes", f"a code is outside [0, {self._k})")
        return self._enc.adc_search(codes, q, int(topk))

    def decode(self, codes) -> np.ndarray:
        """(n, m) codes -> (n, dim) float32 centroids (un-rounded); row blocks over the quantizer's devices"""
        n = int(np.asarray(codes).size) // max(1, self._m)
        return self._batch_encoder(n).decode(codes)

    def dequantize_batch(self, Q) -> np.ndarray:
        """(n, dim) float16 -> (n, dim) float32, row i == dequantize(Q[i]) (src/pq.rs:201-209 for a batch)"""
        Q = np.ascontiguousarray(Q, dtype=np.float16)
        if Q.ndim != 2:
            raise ValueError("expected a 2D array (n, dim)")
        if Q.shape[1] != self._dim:
            raise DimensionMismatch(self._dim, Q.shape[1])
        enc = self._batch_encoder(Q.shape[0])
        return enc.dequantize_f16(Q) if enc is self._menc else _lib.dequantize_f16(Q)
