"""ctypes front-end of the CPU oracle (oracle/vq_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: importable only from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg (see vq_oracle.h).
The package ``vq_amd`` must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SQUARED_EUCLIDEAN, EUCLIDEAN, MANHATTAN, COSINE, COSINE_UNCLAMPED = 0, 1, 2, 3, 4
METRIC_IDS = {
    "squared_euclidean": SQUARED_EUCLIDEAN,
    "euclidean": EUCLIDEAN,
    "manhattan": MANHATTAN,
    "cosine": COSINE,
}

OK, ERR_EMPTY_INPUT, ERR_DIMENSION_MISMATCH, ERR_INVALID_PARAMETER = 0, 1, 2, 3
ERR_RESEED_EXHAUSTED, ERR_REFERENCE_PANICS, ERR_ALLOC = 4, 5, 6


class OracleError(RuntimeError):
    def __init__(self, code: int, where: str):
        super().__init__(f"oracle {where} failed with status {code}")
        self.code = code


def build(native: bool = False) -> str:
    """(Re)build the shared library with the committed Makefile; returns its path."""
    name = "libvq_oracle_native.so" if native else "libvq_oracle.so"
    override = os.environ.get("VQ_ORACLE_LIB")  # tools/run_asan.sh: the sanitizer build of the same source
    if override and not native:
        return override
    path = os.path.join(_HERE, name)
    src = [os.path.join(_HERE, f) for f in ("vq_oracle.c", "vq_oracle.h", "Makefile")]
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in src
    )
    if stale:
        subprocess.run(
            ["make", "-C", _HERE, "native" if native else "all"],
            check=True,
            stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT,
        )
    return path


_f32p = C.POINTER(C.c_float)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_szp = C.POINTER(C.c_size_t)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


class Oracle:
    """Thin wrapper; every method takes/returns numpy arrays."""

    def __init__(self, native: bool = False):
        self.path = build(native)
        L = self.lib = C.CDLL(self.path)
        sz = C.c_size_t
        L.vqo_dot.restype = C.c_float
        L.vqo_dot.argtypes = [_f32p, _f32p, sz]
        L.vqo_norm.restype = C.c_float
        L.vqo_norm.argtypes = [_f32p, sz]
        L.vqo_distance2.restype = C.c_float
        L.vqo_distance2.argtypes = [_f32p, _f32p, sz]
        L.vqo_distance.restype = C.c_float
        L.vqo_distance.argtypes = [C.c_int, _f32p, _f32p, sz]
        L.vqo_mean_vector.restype = C.c_int
        L.vqo_mean_vector.argtypes = [_f32p, sz, sz, sz, _f32p]
        L.vqo_f32_to_f16.restype = C.c_uint16
        L.vqo_f32_to_f16.argtypes = [C.c_float]
        L.vqo_f16_to_f32.restype = C.c_float
        L.vqo_f16_to_f32.argtypes = [C.c_uint16]
        L.vqo_find_nearest.restype = sz
        L.vqo_find_nearest.argtypes = [_f32p, _f32p, sz, sz]
        L.vqo_find_nearest_metric.restype = sz
        L.vqo_find_nearest_metric.argtypes = [C.c_int, _f32p, _f32p, sz, sz]
        L.vqo_lloyd_step.restype = C.c_int
        L.vqo_lloyd_step.argtypes = [_f32p, sz, sz, sz, sz, _f32p, _u32p, _u32p,
                                     C.POINTER(C.c_int), C.c_int]
        L.vqo_lloyd.restype = C.c_int
        L.vqo_lloyd.argtypes = [_f32p, sz, sz, sz, sz, sz, _u64p, _u64p, sz, _f32p, _szp, _szp,
                                C.c_int]
        L.vqo_pq_fit.restype = C.c_int
        L.vqo_pq_fit.argtypes = [_f32p, sz, sz, sz, sz, sz, _u64p, _u64p, sz, _f32p, _szp,
                                 C.c_int]
        L.vqo_adc_search.restype = C.c_int
        L.vqo_adc_search.argtypes = [C.c_int, _f32p, sz, sz, sz, C.POINTER(C.c_uint8), sz, _f32p, sz, sz, _u32p, _f32p]
        L.vqo_adc_search16.restype = C.c_int
        L.vqo_adc_search16.argtypes = [C.c_int, _f32p, sz, sz, sz, C.POINTER(C.c_uint16), sz, _f32p, sz, sz, _u32p, _f32p]
        L.vqo_pq_encode.restype = C.c_int
        L.vqo_pq_encode.argtypes = [C.c_int, _f32p, sz, sz, sz, sz, _f32p, _u32p, _u16p, C.c_int]
        L.vqo_tsvq_build.restype = C.c_int
        L.vqo_tsvq_build.argtypes = [_f32p, sz, sz, sz, sz, _f32p, _i32p, _i32p, _i32p, _u64p]
        L.vqo_tsvq_encode.restype = C.c_int
        L.vqo_tsvq_encode.argtypes = [C.c_int, _f32p, sz, sz, _f32p, _i32p, _i32p, _i32p, _u16p,
                                      C.c_int]
        L.vqo_max_threads.restype = C.c_int

    # -- primitives ------------------------------------------------------------------
    @staticmethod
    def _f32(a):
        return np.ascontiguousarray(a, dtype=np.float32)

    def max_threads(self) -> int:
        return int(self.lib.vqo_max_threads())

    def dot(self, a, b) -> np.float32:
        a, b = self._f32(a), self._f32(b)
        return np.float32(self.lib.vqo_dot(_ptr(a, _f32p), _ptr(b, _f32p), a.size))

    def norm(self, a) -> np.float32:
        a = self._f32(a)
        return np.float32(self.lib.vqo_norm(_ptr(a, _f32p), a.size))

    def distance2(self, a, b) -> np.float32:
        a, b = self._f32(a), self._f32(b)
        return np.float32(self.lib.vqo_distance2(_ptr(a, _f32p), _ptr(b, _f32p), a.size))

    def distance(self, metric: int, a, b) -> np.float32:
        a, b = self._f32(a), self._f32(b)
        if a.size != b.size:
            raise OracleError(ERR_DIMENSION_MISMATCH, "distance")
        return np.float32(self.lib.vqo_distance(metric, _ptr(a, _f32p), _ptr(b, _f32p), a.size))

    def mean_vector(self, rows) -> np.ndarray:
        rows = self._f32(rows)
        n, d = rows.shape
        out = np.empty(d, np.float32)
        rc = self.lib.vqo_mean_vector(_ptr(rows, _f32p), n, d, d, _ptr(out, _f32p))
        if rc:
            raise OracleError(rc, "mean_vector")
        return out

    def f32_to_f16_bits(self, x) -> np.ndarray:
        x = self._f32(x).ravel()
        return np.array([self.lib.vqo_f32_to_f16(float(v)) for v in x], dtype=np.uint16)

    def f16_bits_to_f32(self, h) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.uint16).ravel()
        return np.array([self.lib.vqo_f16_to_f32(int(v)) for v in h], dtype=np.float32)

    def find_nearest(self, x, centroids) -> int:
        x, c = self._f32(x), self._f32(centroids)
        k, sd = c.shape
        return int(self.lib.vqo_find_nearest(_ptr(x, _f32p), _ptr(c, _f32p), k, sd))

    def find_nearest_metric(self, metric: int, x, centroids) -> int:
        x, c = self._f32(x), self._f32(centroids)
        k, sd = c.shape
        return int(self.lib.vqo_find_nearest_metric(metric, _ptr(x, _f32p), _ptr(c, _f32p), k, sd))

    # -- Lloyd -----------------------------------------------------------------------
    def lloyd_step(self, data, centroids, threads: int = 1):
        """data [n][sd] (may be a strided view of a wider matrix); centroids [k][sd].
        Returns (new_centroids, assign u32[n], counts u32[k], changed bool)."""
        data = np.asarray(data, dtype=np.float32)
        assert data.ndim == 2 and (data.shape[0] == 0 or data.strides[1] == 4)
        n, sd = data.shape
        stride = data.strides[0] // 4 if n else sd
        cent = np.array(centroids, dtype=np.float32, order="C", copy=True)
        k = cent.shape[0]
        assign = np.empty(n, np.uint32)
        counts = np.empty(k, np.uint32)
        changed = C.c_int(0)
        rc = self.lib.vqo_lloyd_step(data.ctypes.data_as(_f32p), n, stride, sd, k,
                                     _ptr(cent, _f32p), _ptr(assign, _u32p), _ptr(counts, _u32p),
                                     C.byref(changed), threads)
        if rc:
            raise OracleError(rc, "lloyd_step")
        return cent, assign, counts, bool(changed.value)

    def lloyd(self, data, k: int, max_iters: int, init_rows, reseed_rows=(), threads: int = 1):
        """Returns (centroids [k][sd], iters, reseeds_used)."""
        data = np.asarray(data, dtype=np.float32)
        assert data.ndim == 2 and (data.shape[0] == 0 or data.strides[1] == 4)
        n, sd = data.shape
        stride = data.strides[0] // 4 if n else sd
        init = np.ascontiguousarray(init_rows, dtype=np.uint64)
        rs = np.ascontiguousarray(reseed_rows, dtype=np.uint64)
        out = np.zeros((k, sd), np.float32)
        iters, used = C.c_size_t(0), C.c_size_t(0)
        rc = self.lib.vqo_lloyd(data.ctypes.data_as(_f32p), n, stride, sd, k, max_iters,
                                _ptr(init, _u64p), _ptr(rs, _u64p) if rs.size else None, rs.size,
                                _ptr(out, _f32p), C.byref(iters), C.byref(used), threads)
        if rc:
            raise OracleError(rc, "lloyd")
        return out, int(iters.value), int(used.value)

    def pq_fit(self, rows, m: int, k: int, max_iters: int, init_rows, reseed_rows=None,
               threads: int = 1):
        """rows [n][d]; init_rows [m][k]; reseed_rows [m][r] or None.
        Returns (codebooks [m][k][d/m], iters [m])."""
        rows = self._f32(rows)
        n, d = rows.shape if rows.ndim == 2 else (0, 0)
        init = np.ascontiguousarray(init_rows, dtype=np.uint64)
        rs = None if reseed_rows is None else np.ascontiguousarray(reseed_rows, dtype=np.uint64)
        sd = d // m if m and d % m == 0 else 0
        out = np.zeros((m, k, max(sd, 1)), np.float32)
        iters = (C.c_size_t * max(m, 1))()
        rc = self.lib.vqo_pq_fit(_ptr(rows, _f32p), n, d, m, k, max_iters, _ptr(init, _u64p),
                                 _ptr(rs, _u64p), 0 if rs is None else rs.shape[1],
                                 _ptr(out, _f32p), iters, threads)
        if rc:
            raise OracleError(rc, "pq_fit")
        return out, np.array(list(iters)[:m], dtype=np.int64)

    def pq_encode(self, metric: int, rows, codebooks, want_f16: bool = True, threads: int = 1):
        """rows [n][d]; codebooks [m][k][sd].  Returns (codes u32 [n][m], f16 bits u16 [n][d])."""
        rows = self._f32(rows)
        cb = self._f32(codebooks)
        n, d = rows.shape
        m, k, sd = cb.shape
        assert m * sd == d
        codes = np.empty((n, m), np.uint32)
        f16 = np.empty((n, d), np.uint16) if want_f16 else None
        rc = self.lib.vqo_pq_encode(metric, _ptr(rows, _f32p), n, d, m, k, _ptr(cb, _f32p),
                                    _ptr(codes, _u32p), _ptr(f16, _u16p), threads)
        if rc:
            raise OracleError(rc, "pq_encode")
        return codes, f16

    def adc_search(self, metric: int, codebooks, codes, queries, topk: int):
        """semantics of the code-based search (no reference counterpart): (idx u32, dist f32) [nq][topk]"""
        cb = self._f32(codebooks)
        m, k, sd = cb.shape
        wide = k > 256  # two-byte codes, like the library
        codes = np.ascontiguousarray(codes, np.uint16 if wide else np.uint8)
        q = self._f32(queries)
        n, nq = codes.shape[0], q.shape[0]
        idx = np.empty((nq, topk), np.uint32)
        dist = np.empty((nq, topk), np.float32)
        fn = self.lib.vqo_adc_search16 if wide else self.lib.vqo_adc_search
        cptr = codes.ctypes.data_as(C.POINTER(C.c_uint16 if wide else C.c_uint8))
        rc = fn(metric, _ptr(cb, _f32p), m, k, sd, cptr, n, _ptr(q, _f32p), nq, topk, _ptr(idx, _u32p), _ptr(dist, _f32p))
        if rc:
            raise OracleError(rc, "adc_search")
        return idx, dist

    # -- TSVQ ------------------------------------------------------------------------
    def tsvq_build(self, rows, max_depth: int):
        """Returns dict(centroids [nodes][d], left, right (int32, -1 = none), node_rows)."""
        rows = self._f32(rows)
        if rows.ndim != 2 or rows.shape[0] == 0:
            raise OracleError(ERR_EMPTY_INPUT, "tsvq_build")
        n, d = rows.shape
        cap = min(2 ** (max_depth + 1) - 1, 2 * n - 1) if max_depth < 40 else 2 * n - 1
        cent = np.zeros((cap, d), np.float32)
        left = np.full(cap, -1, np.int32)
        right = np.full(cap, -1, np.int32)
        nrows = np.zeros(cap, np.uint64)
        nn = C.c_int32(0)
        rc = self.lib.vqo_tsvq_build(_ptr(rows, _f32p), n, d, max_depth, cap, _ptr(cent, _f32p),
                                     _ptr(left, _i32p), _ptr(right, _i32p), C.byref(nn),
                                     _ptr(nrows, _u64p))
        if rc:
            raise OracleError(rc, "tsvq_build")
        k = nn.value
        return dict(centroids=cent[:k].copy(), left=left[:k].copy(), right=right[:k].copy(),
                    node_rows=nrows[:k].copy())

    def tsvq_encode(self, metric: int, rows, tree, want_f16: bool = True, threads: int = 1):
        rows = self._f32(rows)
        n, d = rows.shape
        cent = self._f32(tree["centroids"])
        left = np.ascontiguousarray(tree["left"], dtype=np.int32)
        right = np.ascontiguousarray(tree["right"], dtype=np.int32)
        leaf = np.empty(n, np.int32)
        f16 = np.empty((n, d), np.uint16) if want_f16 else None
        rc = self.lib.vqo_tsvq_encode(metric, _ptr(rows, _f32p), n, d, _ptr(cent, _f32p),
                                      _ptr(left, _i32p), _ptr(right, _i32p), _ptr(leaf, _i32p),
                                      _ptr(f16, _u16p), threads)
        if rc:
            raise OracleError(rc, "tsvq_encode")
        return leaf, f16


_default = None


def get(native: bool = False) -> Oracle:
    global _default
    if native:
        return Oracle(native=True)
    if _default is None:
        _default = Oracle()
    return _default
