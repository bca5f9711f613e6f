"""``Distance`` -- host mirror of pyvq.Distance (reference pyvq/src/distance.rs:17-108) and of
the Rust enum it wraps (src/core/distance.rs:8-28)."""
from __future__ import annotations

import numpy as np

from . import _lib
from .errors import DimensionMismatch

_NAMES = {
    "euclidean": _lib.EUCLIDEAN,
    "squaredeuclidean": _lib.SQUARED_EUCLIDEAN,
    "squared_euclidean": _lib.SQUARED_EUCLIDEAN,
    "cosine": _lib.COSINE,
    "cosine_distance": _lib.COSINE,
    "manhattan": _lib.MANHATTAN,
    # opt-in, UNPINNED (include/vqhip.h): the reference's `simd` build, 1 - similarity without EPSILON rule or clamp
    "cosine_unclamped": _lib.COSINE_UNCLAMPED,
    "cosine_simd": _lib.COSINE_UNCLAMPED,
}
_CANON = {
    _lib.EUCLIDEAN: "euclidean",
    _lib.SQUARED_EUCLIDEAN: "squared_euclidean",
    _lib.MANHATTAN: "manhattan",
    _lib.COSINE: "cosine",
    _lib.COSINE_UNCLAMPED: "cosine_unclamped",
}


class Distance:
    """Distance metric selector.  ``Distance("euclidean")`` or ``Distance.euclidean()``."""

    def __init__(self, metric: str):
        key = str(metric).lower()
        if key not in _NAMES:
            # message of pyvq/src/distance.rs:40-42
            raise ValueError(
                "Invalid distance metric. Choose from: euclidean, squared_euclidean, cosine, manhattan")
        self.metric = _NAMES[key]

    @staticmethod
    def euclidean() -> "Distance":
        return Distance("euclidean")

    @staticmethod
    def squared_euclidean() -> "Distance":
        return Distance("squared_euclidean")

    @staticmethod
    def manhattan() -> "Distance":
        return Distance("manhattan")

    @staticmethod
    def cosine() -> "Distance":
        return Distance("cosine")

    @staticmethod
    def cosine_unclamped() -> "Distance":
        """The reference's `simd`-feature cosine, ``1 - similarity`` with no EPSILON rule and no clamp
        (src/core/distance.rs:97-105).  Opt-in and unpinned: hsdlib's summation order is unknown."""
        return Distance("cosine_unclamped")

    def name(self) -> str:
        """``Distance::name`` (src/core/distance.rs:21-28)"""
        return _CANON[self.metric]

    def compute(self, a, b) -> float:
        """Distance between two vectors (``Distance::compute``, src/core/distance.rs:48-64).

        Evaluated on the device as a 1-row, 1-centroid problem through the exact kernel is
        pointless; the single-pair value is obtained from the batch distance entry point."""
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        b = np.ascontiguousarray(b, dtype=np.float32).ravel()
        if a.size != b.size:
            raise DimensionMismatch(a.size, b.size)
        from ._pairwise import pairwise_distance

        return float(pairwise_distance(self.metric, a[None, :], b[None, :])[0])

    def __repr__(self) -> str:
        return f"Distance('{_CANON[self.metric]}')"

    def __eq__(self, other) -> bool:
        return isinstance(other, Distance) and other.metric == self.metric

    def __hash__(self) -> int:
        return hash(self.metric)
