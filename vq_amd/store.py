"""Code-based storage of a product-quantized set.

The reference's `quantize` hands back the f16 reconstruction (2 bytes per dimension,
src/pq.rs:165-199); the same information is the m one-byte codes per vector plus the codebooks
(C2: 8 bytes instead of 256 per vector).  `PQIndex` keeps exactly that and reproduces the
reference outputs from it: `reconstruct_f16()` == `quantize` of the encoded rows, bit for bit.

File layout (little endian), stable and trivially readable from C / Rust:

    0   8   magic  b"VQPQIDX1"
    8   4   u32    metric (0 squared_euclidean, 1 euclidean, 2 manhattan, 3 cosine)
    12  4   u32    dim
    16  4   u32    m
    20  4   u32    k
    24  8   u64    n
    32  ..  f32    codebooks [m][k][dim/m]
    ..  ..  u8     codes     [n][m]        (k <= 256)
            u16    codes     [n][m]        (256 < k <= 65536, little endian; the library's code width)
"""
from __future__ import annotations

import struct

import numpy as np

from .distance import Distance

MAGIC = b"VQPQIDX1"
_HEADER = struct.Struct("<8sIIIIQ")
# metric ids of include/vqhip.h in id order (4 = the simd build's cosine without the epsilon rule and the clamp)
_METRIC_NAMES = ["squared_euclidean", "euclidean", "manhattan", "cosine", "cosine_unclamped"]


def _code_dtype(k: int):
    return np.dtype(np.uint8) if k <= 256 else np.dtype("<u2")


class PQIndex:
    def __init__(self, codebooks: np.ndarray, codes: np.ndarray, distance: Distance | None = None):
        cb = np.ascontiguousarray(codebooks, dtype=np.float32)
        if cb.ndim != 3:
            raise ValueError("codebooks must have shape (m, k, sub_dim)")
        if cb.shape[1] > 65536:
            raise ValueError("codes are at most two bytes: k <= 65536")
        codes = np.asarray(codes)
        if codes.ndim != 2 or codes.shape[1] != cb.shape[0]:
            raise ValueError(f"codes must have shape (n, {cb.shape[0]})")
        if codes.size and (int(codes.min()) < 0 or int(codes.max()) >= cb.shape[1]):
            raise ValueError("code out of range for the codebooks")
        codes = np.ascontiguousarray(codes, dtype=_code_dtype(cb.shape[1]))
        self.codebooks, self.codes = cb, codes
        self.distance = distance if distance is not None else Distance.euclidean()

    # -- shape ------------------------------------------------------------------------------
    @property
    def m(self) -> int:
        return self.codebooks.shape[0]

    @property
    def k(self) -> int:
        return self.codebooks.shape[1]

    @property
    def dim(self) -> int:
        return self.codebooks.shape[0] * self.codebooks.shape[2]

    def __len__(self) -> int:
        return self.codes.shape[0]

    @property
    def nbytes(self) -> int:
        return _HEADER.size + self.codebooks.nbytes + self.codes.nbytes

    # -- build ------------------------------------------------------------------------------
    @classmethod
    def from_quantizer(cls, pq, X) -> "PQIndex":
        """encode X with a trained `ProductQuantizer` (device path) and keep the codes"""
        return cls(pq.codebooks, pq.encode(X), Distance(pq.distance_metric()))

    # -- reference-shaped outputs from the codes ----------------------------------------------
    def reconstruct(self, rows=None) -> np.ndarray:
        """(n, dim) float32: the selected centroids (host gather; the device path is pq.decode)"""
        codes = self.codes if rows is None else self.codes[rows]
        m, sd = self.m, self.codebooks.shape[2]
        out = np.empty((codes.shape[0], m * sd), np.float32)
        for s in range(m):
            out[:, s * sd:(s + 1) * sd] = self.codebooks[s][codes[:, s]]
        return out

    def reconstruct_f16(self, rows=None) -> np.ndarray:
        """what `ProductQuantizer::quantize` returned for these rows (f16, RNE; src/pq.rs:192-196)"""
        return self.reconstruct(rows).astype(np.float16)

    def search(self, queries, topk: int = 10):
        """top-k stored rows per query by asymmetric distance (device path; see ProductQuantizer.search)"""
        from . import _lib
        from .errors import DimensionMismatch, InvalidParameter

        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise DimensionMismatch(self.dim, q.shape[1])
        if not 1 <= topk <= min(len(self), 1024):
            raise InvalidParameter("topk", f"must be between 1 and min(n, 1024), got {topk}")
        if self.distance.metric in (_lib.COSINE, _lib.COSINE_UNCLAMPED):
            raise InvalidParameter("distance", "cosine distance is not a sum over subspaces: no ADC form")
        # the codes go to the device ONCE (checked against k there); later searches only send the queries
        enc = getattr(self, "_enc", None)
        if enc is None:
            enc = _lib.PQEncoder(self.codebooks, self.distance.metric)
            enc.adc_set_codes(np.asarray(self.codes))
            self._enc = enc
        return enc.adc_search(None, q, int(topk))

    # -- file -------------------------------------------------------------------------------
    def save(self, path) -> None:
        with open(path, "wb") as f:
            f.write(_HEADER.pack(MAGIC, self.distance.metric, self.dim, self.m, self.k, len(self)))
            f.write(self.codebooks.tobytes())
            f.write(self.codes.tobytes())

    @classmethod
    def load(cls, path, mmap_codes: bool = False) -> "PQIndex":
        with open(path, "rb") as f:
            head = f.read(_HEADER.size)
            if len(head) != _HEADER.size:
                raise ValueError("truncated index header")
            magic, metric, dim, m, k, n = _HEADER.unpack(head)
            if magic != MAGIC:
                raise ValueError("not a VQPQIDX1 file")
            if m == 0 or k == 0 or k > 65536 or dim == 0 or dim % m != 0 or metric >= len(_METRIC_NAMES):
                raise ValueError("corrupt index header")
            sd = dim // m
            cb = np.frombuffer(f.read(m * k * sd * 4), dtype="<f4")
            if cb.size != m * k * sd:
                raise ValueError("truncated codebooks")
            off = f.tell()
            cdt = _code_dtype(k)
            if mmap_codes:
                codes = np.memmap(path, dtype=cdt, mode="r", offset=off, shape=(n, m))
            else:
                raw = f.read(n * m * cdt.itemsize)
                if len(raw) != n * m * cdt.itemsize:
                    raise ValueError("truncated codes")
                codes = np.frombuffer(raw, dtype=cdt).reshape(n, m)
        # k_adc_scan indexes the per-query LDS tables by code: a corrupt file must not get that far
        for r0 in range(0, n, 1 << 22):
            if codes[r0:r0 + (1 << 22)].size and int(codes[r0:r0 + (1 << 22)].max()) >= k:
                raise ValueError(f"corrupt index: a code is outside [0, {k})")
        self = cls.__new__(cls)
        self.codebooks = cb.reshape(m, k, sd).astype(np.float32)
        self.codes = codes
        self.distance = Distance(_METRIC_NAMES[metric])
        return self
