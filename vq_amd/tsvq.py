"""``TSVQ`` -- host mirror of the reference's tree-structured quantizer.

Same constructor, properties, ``repr`` and error text as pyvq.TSVQ (reference
pyvq/src/tsvq.rs:30-121) / ``TSVQ::new`` (src/tsvq.rs:195-223).  The tree build
(``TSVQNode::build``, src/tsvq.rs:31-115) and the descent (``find_leaf``, 117-132) run on the
MI355X through libvqhip; the tree is kept as flat pre-order arrays.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _arena, _lib
from .distance import Distance
from .errors import DimensionMismatch
from .pq import _as_training_matrix


class _TsvqHandle(_lib.Handle):
    _destroy = "vqhip_tsvq_destroy"


def build_tree(ds: "_lib.Dataset", max_depth: int):
    """(centroids [nodes][d], left, right) in pre-order; child index -1 = none."""
    n, d = ds.n, ds.d
    cap = min(2 ** (max_depth + 1) - 1, 2 * n - 1) if max_depth < 40 else 2 * n - 1
    cent = np.zeros((cap, d), np.float32)
    left = np.full(cap, -1, np.int32)
    right = np.full(cap, -1, np.int32)
    nn = C.c_int32(0)
    lib = _lib.load()
    _lib.check(lib.vqhip_tsvq_build(ds.raw, int(max_depth), int(cap), _lib.ptr(cent, _lib._f32p),
                                    _lib.ptr(left, _lib._i32p), _lib.ptr(right, _lib._i32p), C.byref(nn)))
    k = nn.value
    return cent[:k].copy(), left[:k].copy(), right[:k].copy()


class TSVQ:
    """Tree-structured vector quantizer.

    Args (pyvq/src/tsvq.rs:41-42): training_data (n, dim) float32; max_depth; distance=None
    (-> Euclidean)."""

    def __init__(self, training_data, max_depth: int, distance: Distance | None = None, *, devices=None):
        """devices (not in pyvq): the GPUs batch encodes are split over in row blocks -- None: every visible device; an
        int n: 0..n-1; a sequence of ids.  The BUILD runs on one device (the calling thread's current one): its column
        sums are the reference's sequential chains over a node's rows (replicas only); the tree (260 KB
        at depth 8, d = 128) is then replicated and every device descends its own rows -- leaves and f16 rows are the
        same bits whatever the device list."""
        X = _as_training_matrix(training_data)
        self._dim = X.shape[1]
        self._distance = distance if distance is not None else Distance.euclidean()
        ds = _lib.Dataset.from_host(X)
        try:
            self._centroids, self._left, self._right = build_tree(ds, int(max_depth))
        finally:
            ds.close()
        self._make_encoder(devices)

    @classmethod
    def from_tree(cls, centroids, left, right, distance: Distance | None = None, *, devices=None) -> "TSVQ":
        self = cls.__new__(cls)
        self._centroids = np.ascontiguousarray(centroids, dtype=np.float32)
        self._left = np.ascontiguousarray(left, dtype=np.int32)
        self._right = np.ascontiguousarray(right, dtype=np.int32)
        self._dim = self._centroids.shape[1]
        self._distance = distance if distance is not None else Distance.euclidean()
        self._make_encoder(devices)
        return self

    def _make_encoder(self, devices=None):
        dv = _lib._devices(devices)
        # the multi-device encoder (a worker thread and a replica of the tree per device slot) is made by the first batch
        # large enough to use it: a TSVQ that only ever quantizes single vectors starts no threads
        self._menc = None
        self._menc_lock = threading.Lock()
        self._dv = dv
        self.devices = [int(x) for x in dv]
        h = C.c_void_p()
        lib = _lib.load()
        _lib.check(lib.vqhip_tsvq_create(_lib.ptr(self._centroids, _lib._f32p), _lib.ptr(self._left, _lib._i32p),
                                         _lib.ptr(self._right, _lib._i32p), self._centroids.shape[0], self._dim,
                                         self._distance.metric, C.byref(h)))
        self._enc = _TsvqHandle(h)

    # -- reference surface ----------------------------------------------------------------
    def quantize(self, vector) -> np.ndarray:
        """float32 (dim,) -> float16 (dim,): the leaf centroid (src/tsvq.rs:239-255)"""
        v = np.ascontiguousarray(vector, dtype=np.float32).ravel()
        if v.size != self._dim:
            raise DimensionMismatch(self._dim, v.size)
        return self.quantize_batch(v[None, :])[0]

    def dequantize(self, codes) -> np.ndarray:
        q = np.ascontiguousarray(codes, dtype=np.float16).ravel()
        if q.size != self._dim:
            raise DimensionMismatch(self._dim, q.size)
        return _lib.dequantize_f16(q)

    @property
    def dim(self) -> int:
        return self._dim

    def distance_metric(self) -> str:
        return self._distance.name()

    def __repr__(self) -> str:  # pyvq/src/tsvq.rs:118-120
        return f"TSVQ(dim={self._dim})"

    # -- batch additions ---------------------------------------------------------------------
    @property
    def tree(self):
        return self._centroids, self._left, self._right

    def _encode(self, X, want_leaf: bool, want_f16: bool):
        X = np.ascontiguousarray(X, dtype=np.float32)
        if X.ndim != 2:
            raise ValueError("expected a 2D array (n, dim)")
        if X.shape[1] != self._dim:
            raise DimensionMismatch(self._dim, X.shape[1])
        n = X.shape[0]
        leaf = _arena.fresh((n,), np.int32) if want_leaf else None  # (large results: recycled buffers, vq_amd/_arena.py)
        f16 = _arena.fresh((n, self._dim), np.uint16) if want_f16 else None
        self._last_multi = False
        if n and self._multi(n) is not None:
            self._menc.encode(X, leaf, f16)
            self._last_multi = True
        elif n:
            _lib.check(_lib.load().vqhip_tsvq_encode(self._enc.raw, _lib.ptr(X, _lib._f32p), n,
                                                     _lib.ptr(leaf, _lib._i32p), _lib.ptr(f16, _lib._u16p)))
        return leaf, (None if f16 is None else f16.view(np.float16))

    def _multi(self, n: int):
        """the row-block encoder over the device list for a batch of n rows, None for batches (or device lists) too small"""
        if self._dv.size < 2 or n < 65536 * self._dv.size:
            return None
        if self._menc is None:
            with self._menc_lock:
                if self._menc is None:
                    self._menc = _lib.MTSVQ(self._centroids, self._left, self._right, self._distance.metric, self._dv)
        return self._menc

    def quantize_batch(self, X) -> np.ndarray:
        return self._encode(X, False, True)[1]

    def last_encode_stats(self):
        """(screened descent used?, rows finished by the exact continuation) of the last batch"""
        if getattr(self, "_last_multi", False):
            return self._menc.last_stats()
        scr, und = C.c_int(0), C.c_uint64(0)
        _lib.check(_lib.load().vqhip_tsvq_last_stats(self._enc.raw, C.byref(scr), C.byref(und)))
        return bool(scr.value), int(und.value)

    def leaf_ids(self, X) -> np.ndarray:
        """node index (pre-order) of the leaf each row descends to"""
        return self._encode(X, True, False)[0]

    def dequantize_batch(self, Q) -> np.ndarray:
        """(n, dim) float16 -> (n, dim) float32, row i == dequantize(Q[i]) (src/tsvq.rs:257-265 for a batch)"""
        Q = np.ascontiguousarray(Q, dtype=np.float16)
        if Q.ndim != 2:
            raise ValueError("expected a 2D array (n, dim)")
        if Q.shape[1] != self._dim:
            raise DimensionMismatch(self._dim, Q.shape[1])
        menc = self._multi(Q.shape[0])
        return menc.dequantize_f16(Q) if menc is not None else _lib.dequantize_f16(Q)
