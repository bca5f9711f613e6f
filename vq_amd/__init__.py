"""vq_amd -- MI355X (gfx950) back end for the k-means codebook-training and nearest-centroid
encode path of CogitatorTech/vq, behind the reference's own Quantizer interface.

The compute path is libvqhip.so (hand-written HIP for CDNA4, C ABI in include/vqhip.h).
Importing the package does not need a GPU; using any quantizer does, and fails loudly
(``FfiError``) otherwise -- there is no CPU fallback.
"""
from .distance import Distance
from .errors import (DimensionMismatch, EmptyInput, FfiError, InvalidData, InvalidParameter,
                     VqError)
from .pq import ProductQuantizer, fit_codebooks
from .tsvq import TSVQ

__all__ = [
    "Distance", "ProductQuantizer", "TSVQ", "fit_codebooks", "VqError", "DimensionMismatch", "EmptyInput",
    "InvalidParameter", "InvalidData", "FfiError", "get_simd_backend",
]


def get_simd_backend() -> str:
    """pyvq.get_simd_backend analogue (pyvq/src/lib.rs:19-21): names the active back end."""
    from . import _lib

    return _lib.backend()
