"""Batch pairwise distances on the device (vqhip_distance_batch)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def pairwise_distance(metric: int, a, b) -> np.ndarray:
    """out[i] = metric(a[i], b[i]) for (n, d) float32 arrays, evaluated in the reference's
    arithmetic (src/core/distance.rs:48-64)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape and a.ndim == 2
    n, d = a.shape
    out = np.empty(n, np.float32)
    _lib.check(_lib.load().vqhip_distance_batch(metric, _lib.ptr(a, _lib._f32p), _lib.ptr(b, _lib._f32p),
                                                n, d, _lib.ptr(out, _lib._f32p)))
    return out
