"""ctypes binding of libvqhip.so (include/vqhip.h).

The shared library is built in-tree (``vq_amd/libvqhip.so``, see ``__graft_entry__.build``)
and is the ONLY compute back end: if it is missing or fails to load this module raises --
there is no eager/numpy fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _arena

from .errors import FfiError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQHIP_LIB_PATH", os.path.join(_HERE, "libvqhip.so"))

OK = 0
ERR_NULL_PTR, ERR_INVALID_INPUT, ERR_NO_DEVICE, ERR_RUNTIME, ERR_UNSUPPORTED, ERR_FAILURE = (
    -1, -3, -4, -5, -6, -99)

SQUARED_EUCLIDEAN, EUCLIDEAN, MANHATTAN, COSINE, COSINE_UNCLAMPED = 0, 1, 2, 3, 4
ENGINE_AUTO, ENGINE_EXACT, ENGINE_MFMA, ENGINE_MFMA_BF16 = 0, 1, 2, 3

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)

# every symbol include/vqhip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vqhip_backend": (C.c_char_p, []),
    "vqhip_last_error": (C.c_char_p, []),
    "vqhip_device_count": (C.c_int, []),
    "vqhip_set_device": (C.c_int, [C.c_int]),
    "vqhip_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "vqhip_set_stream": (C.c_int, [_vp]),
    "vqhip_synchronize": (C.c_int, []),
    "vqhip_last_assign_stats": (C.c_int, [_u64p, C.POINTER(C.c_int)]),
    "vqhip_xfer_lane_calls": (C.c_int, [C.POINTER(C.c_uint64)]),
    "vqhip_set_profiling": (C.c_int, [C.c_int]),
    "vqhip_profile_collect": (C.c_int, [_u32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vqhip_memcpy_device": (C.c_int, [_vp, _vp, C.c_uint64]),
    "vqhip_code_bytes": (C.c_uint32, [C.c_uint32]),
    "vqhip_dataset_from_host": (C.c_int, [_f32p, C.c_uint64, C.c_uint32, _vpp]),
    "vqhip_dataset_from_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vpp]),
    "vqhip_dataset_synthetic": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _vpp]),
    "vqhip_dataset_info": (C.c_int, [_vp, _u64p, _u32p, _vpp]),
    "vqhip_dataset_read": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _f32p]),
    "vqhip_dataset_destroy": (C.c_int, [_vp]),
    "vqhip_synth_uniform_host": (C.c_int, [_f32p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]),
    "vqhip_kmeans_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vpp]),
    "vqhip_kmeans_destroy": (C.c_int, [_vp]),
    "vqhip_kmeans_set_centroids": (C.c_int, [_vp, _f32p]),
    "vqhip_kmeans_init_from_rows": (C.c_int, [_vp, _u64p]),
    "vqhip_kmeans_get_centroids": (C.c_int, [_vp, _f32p]),
    "vqhip_kmeans_set_active": (C.c_int, [_vp, _u8p]),
    "vqhip_kmeans_get_active": (C.c_int, [_vp, _u8p]),
    "vqhip_kmeans_set_engine": (C.c_int, [_vp, C.c_int]),
    "vqhip_kmeans_set_exact_update": (C.c_int, [_vp, C.c_int]),
    "vqhip_kmeans_step": (C.c_int, [_vp, _u32p, _u8p]),
    "vqhip_kmeans_run": (C.c_int, [_vp, C.c_uint32, _u32p, _u32p, _u8p, C.POINTER(C.c_int)]),
    "vqhip_kmeans_run_sharded": (C.c_int, [_vp, _vp, C.c_uint32, _u32p, _u32p, _u8p, C.POINTER(C.c_int)]),
    "vqhip_kmeans_accumulate": (C.c_int, [_vp]),
    "vqhip_kmeans_partials": (C.c_int, [_vp, _vpp, _u64p]),
    "vqhip_kmeans_finalize": (C.c_int, [_vp, _u32p, _u8p]),
    "vqhip_comm_unique_id": (C.c_int, [_u8p]),
    "vqhip_comm_create": (C.c_int, [_u8p, C.c_int, C.c_int, _vpp]),
    "vqhip_comm_adopt": (C.c_int, [_vp, _vpp]),
    "vqhip_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vqhip_comm_destroy": (C.c_int, [_vp]),
    "vqhip_kmeans_allreduce": (C.c_int, [_vp, _vp]),
    "vqhip_kmeans_step_sharded": (C.c_int, [_vp, _vp, _u32p, _u8p]),
    "vqhip_kmeans_init_from_global_rows": (C.c_int, [_vp, _vp, _u64p, C.c_uint64]),
    "vqhip_kmeans_patch_from_global_row": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64]),
    "vqhip_kmeans_gather_owned_rows": (C.c_int, [_vp, _u64p, C.c_uint64, _u32p]),
    "vqhip_kmeans_patch_centroid": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _f32p]),
    "vqhip_kmeans_patch_from_row": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint64]),
    "vqhip_kmeans_get_assignments": (C.c_int, [_vp, _u8p]),
    "vqhip_pq_encoder_create": (C.c_int, [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _vpp]),
    "vqhip_pq_encoder_destroy": (C.c_int, [_vp]),
    "vqhip_pq_encoder_set_engine": (C.c_int, [_vp, C.c_int]),
    "vqhip_pq_encode": (C.c_int, [_vp, _f32p, C.c_uint64, _u8p, _u16p]),
    "vqhip_pq_encode_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "vqhip_dequantize_f16": (C.c_int, [_u16p, C.c_uint64, _f32p]),
    "vqhip_pq_decode": (C.c_int, [_vp, _u8p, C.c_uint64, _f32p]),
    "vqhip_distance_batch": (C.c_int, [C.c_int, _f32p, _f32p, C.c_uint64, C.c_uint32, _f32p]),
    "vqhip_tsvq_build": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _f32p, _i32p, _i32p, _i32p]),
    "vqhip_tsvq_create": (C.c_int, [_f32p, _i32p, _i32p, C.c_uint32, C.c_uint32, C.c_int, _vpp]),
    "vqhip_tsvq_destroy": (C.c_int, [_vp]),
    "vqhip_tsvq_encode": (C.c_int, [_vp, _f32p, C.c_uint64, _i32p, _u16p]),
    "vqhip_tsvq_encode_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "vqhip_pq_adc_search": (C.c_int, [_vp, _u8p, C.c_uint64, _f32p, C.c_uint32, C.c_uint32, _u32p, _f32p]),
    "vqhip_pq_adc_search_device": (C.c_int, [_vp, _vp, C.c_uint64, _f32p, C.c_uint32, C.c_uint32, _u32p, _f32p]),
    "vqhip_pq_adc_last_redone": (C.c_int, [_vp, _u32p]),
    "vqhip_pq_adc_set_codes": (C.c_int, [_vp, _u8p, C.c_uint64]),
    "vqhip_pq_adc_search_resident": (C.c_int, [_vp, _f32p, C.c_uint32, C.c_uint32, _u32p, _f32p]),
    "vqhip_selftest": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "vqhip_mfma_bf16_probe": (C.c_int, [_u16p, _u16p, _f32p, C.c_uint64, _f32p]),
    "vqhip_mfma_bf16_model": (C.c_int, [_u16p, _u16p, _f32p, C.c_uint64, _f32p]),
    "vqhip_mfma_bf16_model_check": (C.c_int, [C.c_uint64, C.c_uint64, _u64p, _u64p]),
    "vqhip_mfma_bf16_model_failures": (C.c_int, [C.c_uint64, C.c_uint64, _u64p, C.c_uint32, _u64p]),
    "vqhip_mfma_bf16_model_case": (C.c_int, [C.c_uint64, C.c_uint64, _u16p, _u16p, _f32p]),
    "vqhip_tsvq_last_stats": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "vqhip_comm_group_create": (C.c_int, [C.c_int, _vpp]),
    "vqhip_comm_create_local": (C.c_int, [_vp, C.c_int, _vpp]),
    "vqhip_comm_group_destroy": (C.c_int, [_vp]),
    "vqhip_comm_kind": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "vqhip_mdataset_from_host": (C.c_int, [_f32p, C.c_uint64, C.c_uint32, _i32p, C.c_int, _vpp]),
    "vqhip_mdataset_synthetic": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, _i32p, C.c_int, _vpp]),
    "vqhip_mdataset_info": (C.c_int, [_vp, _u64p, _u32p, C.POINTER(C.c_int), _u64p]),
    "vqhip_mdataset_destroy": (C.c_int, [_vp]),
    "vqhip_mkmeans_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vpp]),
    "vqhip_mkmeans_destroy": (C.c_int, [_vp]),
    "vqhip_mkmeans_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vqhip_mkmeans_set_engine": (C.c_int, [_vp, C.c_int]),
    "vqhip_mkmeans_set_exact_update": (C.c_int, [_vp, C.c_int]),
    "vqhip_mkmeans_init_from_rows": (C.c_int, [_vp, _u64p]),
    "vqhip_mkmeans_set_centroids": (C.c_int, [_vp, _f32p]),
    "vqhip_mkmeans_get_centroids": (C.c_int, [_vp, _f32p]),
    "vqhip_mkmeans_set_active": (C.c_int, [_vp, _u8p]),
    "vqhip_mkmeans_get_active": (C.c_int, [_vp, _u8p]),
    "vqhip_mkmeans_run": (C.c_int, [_vp, C.c_uint32, _u32p, _u32p, _u8p, C.POINTER(C.c_int)]),
    "vqhip_mkmeans_patch_from_row": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint64]),
    "vqhip_mpq_encoder_create": (C.c_int, [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _i32p, C.c_int, _vpp]),
    "vqhip_mpq_encoder_set_engine": (C.c_int, [_vp, C.c_int]),
    "vqhip_mpq_encode": (C.c_int, [_vp, _f32p, C.c_uint64, _u8p, _u16p]),
    "vqhip_mpq_encoder_destroy": (C.c_int, [_vp]),
    "vqhip_mpq_encode_dataset": (C.c_int, [_vp, _vp, C.c_uint32, _u8p]),
    "vqhip_shard_rows": (C.c_int, [C.c_uint64, C.c_int, C.c_int, _u64p, _u64p]),
    "vqhip_comm_abort": (C.c_int, [_vp]),
    "vqhip_pq_decode_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "vqhip_dequantize_f16_device": (C.c_int, [_vp, C.c_uint64, _vp]),
    "vqhip_mpq_decode": (C.c_int, [_vp, _u8p, C.c_uint64, _f32p]),
    "vqhip_mpq_dequantize_f16": (C.c_int, [_vp, _u16p, C.c_uint64, _f32p]),
    "vqhip_mtsvq_create": (C.c_int, [_f32p, _i32p, _i32p, C.c_uint32, C.c_uint32, C.c_int, _i32p, C.c_int, _vpp]),
    "vqhip_mtsvq_encode": (C.c_int, [_vp, _f32p, C.c_uint64, _i32p, _u16p]),
    "vqhip_mtsvq_dequantize_f16": (C.c_int, [_vp, _u16p, C.c_uint64, _f32p]),
    "vqhip_mtsvq_last_stats": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "vqhip_mtsvq_destroy": (C.c_int, [_vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libvqhip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FfiError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C vq_amd/csrc`).  vq_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64.
    # If libvqhip pulled in /opt/rocm's copy first, torch's later initialisation finds no GPU.
    # Importing torch first makes both resolve the same SONAMEs to the same loaded objects
    # (plumbing only: nothing here calls into torch).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-free hosts use the system runtime
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise FfiError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise FfiError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().vqhip_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != OK:
        raise FfiError(last_error() or f"libvqhip status {rc}", rc)


def ptr(a: np.ndarray | None, ty):
    return None if a is None else a.ctypes.data_as(ty)


def f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def code_dtype(k: int):
    """codes are one byte while k <= 256 and a u16 above (include/vqhip.h, "code width")"""
    return np.uint8 if int(k) <= 256 else np.uint16


def _encode_outputs(n: int, m: int, k: int, dim: int, want_codes: bool, want_f16: bool, out_codes, out_f16):
    """(codes, f16) arrays an encode call fills: fresh ones (large: recycled buffers, vq_amd/_arena.py), or the caller's
    `out_*` after the checks that keep a bad one an FfiError instead of a heap overwrite by the library's threads"""
    codes = f16 = None
    if want_codes or out_codes is not None:
        codes = out_codes if out_codes is not None else _arena.fresh((n, m), code_dtype(k))
        if (not isinstance(codes, np.ndarray) or codes.shape != (n, m) or codes.dtype != code_dtype(k)
                or not codes.flags.c_contiguous or not codes.flags.writeable):
            raise FfiError(f"out_codes must be a writable C-contiguous {np.dtype(code_dtype(k)).name} array of shape ({n}, {m})", ERR_INVALID_INPUT)
    if want_f16 or out_f16 is not None:
        f16 = out_f16 if out_f16 is not None else _arena.fresh((n, dim), np.float16)
        if (not isinstance(f16, np.ndarray) or f16.shape != (n, dim) or f16.dtype.itemsize != 2 or f16.dtype.kind not in "fu"
                or not f16.flags.c_contiguous or not f16.flags.writeable):
            raise FfiError(f"out_f16 must be a writable C-contiguous float16 array of shape ({n}, {dim})", ERR_INVALID_INPUT)
    return codes, f16


class Handle:
    """Owns one opaque library handle and destroys it with the matching *_destroy."""

    _destroy = ""

    def __init__(self, raw: C.c_void_p):
        self.raw = raw

    def close(self):
        if getattr(self, "raw", None):
            getattr(load(), self._destroy)(self.raw)
            self.raw = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass


class Dataset(Handle):
    """Row-major [n][d] f32 matrix resident in HBM (vqhip_dataset)."""

    _destroy = "vqhip_dataset_destroy"

    def __init__(self, raw, n: int, d: int, keepalive=None):
        super().__init__(raw)
        self.n, self.d = int(n), int(d)
        self._keepalive = keepalive

    @classmethod
    def from_host(cls, rows: np.ndarray) -> "Dataset":
        rows = f32c(rows)
        n, d = rows.shape
        h = C.c_void_p()
        check(load().vqhip_dataset_from_host(ptr(rows, _f32p), n, d, C.byref(h)))
        return cls(h, n, d)

    @classmethod
    def from_device(cls, dev_ptr: int, n: int, d: int, keepalive=None) -> "Dataset":
        h = C.c_void_p()
        check(load().vqhip_dataset_from_device(C.c_void_p(dev_ptr), n, d, C.byref(h)))
        return cls(h, n, d, keepalive)

    @classmethod
    def synthetic(cls, n: int, d: int, seed: int, row_offset: int = 0) -> "Dataset":
        h = C.c_void_p()
        check(load().vqhip_dataset_synthetic(n, d, seed, row_offset, C.byref(h)))
        return cls(h, n, d)

    @property
    def device_ptr(self) -> int:
        p = C.c_void_p()
        check(load().vqhip_dataset_info(self.raw, None, None, C.byref(p)))
        return int(p.value)

    def read(self, row0: int = 0, nrows: int | None = None) -> np.ndarray:
        nrows = self.n - row0 if nrows is None else nrows
        out = np.empty((nrows, self.d), np.float32)
        check(load().vqhip_dataset_read(self.raw, row0, nrows, ptr(out, _f32p)))
        return out


def synth_uniform_host(n: int, d: int, seed: int, row_offset: int = 0) -> np.ndarray:
    out = np.empty((n, d), np.float32)
    check(load().vqhip_synth_uniform_host(ptr(out, _f32p), n, d, seed, row_offset))
    return out


class KMeans(Handle):
    """vqhip_kmeans: Lloyd iterations over all m subspaces of a resident dataset."""

    _destroy = "vqhip_kmeans_destroy"

    def __init__(self, ds: Dataset, m: int, k: int):
        h = C.c_void_p()
        check(load().vqhip_kmeans_create(ds.raw, m, k, C.byref(h)))
        super().__init__(h)
        self.ds, self.m, self.k, self.sd = ds, int(m), int(k), ds.d // int(m)

    def set_centroids(self, c):
        c = f32c(c).reshape(self.m, self.k, self.sd)
        check(load().vqhip_kmeans_set_centroids(self.raw, ptr(c, _f32p)))

    def init_from_rows(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.uint64).reshape(self.m, self.k)
        check(load().vqhip_kmeans_init_from_rows(self.raw, ptr(r, _u64p)))

    def get_centroids(self) -> np.ndarray:
        out = np.empty((self.m, self.k, self.sd), np.float32)
        check(load().vqhip_kmeans_get_centroids(self.raw, ptr(out, _f32p)))
        return out

    def set_active(self, active):
        a = np.ascontiguousarray(active, dtype=np.uint8).reshape(self.m)
        check(load().vqhip_kmeans_set_active(self.raw, ptr(a, _u8p)))

    def get_active(self) -> np.ndarray:
        """bool [m]: the library's own active set (``run`` retires converged subspaces on the device)"""
        a = np.zeros(self.m, np.uint8)
        check(load().vqhip_kmeans_get_active(self.raw, ptr(a, _u8p)))
        return a.astype(bool)

    def set_engine(self, engine: int):
        check(load().vqhip_kmeans_set_engine(self.raw, engine))

    def set_exact_update(self, on: bool):
        check(load().vqhip_kmeans_set_exact_update(self.raw, 1 if on else 0))

    def step(self):
        counts = np.empty((self.m, self.k), np.uint32)
        changed = np.empty(self.m, np.uint8)
        check(load().vqhip_kmeans_step(self.raw, ptr(counts, _u32p), ptr(changed, _u8p)))
        return counts, changed.astype(bool)

    def run(self, max_iters: int, comm: "NativeComm | None" = None):
        """up to max_iters iterations, decisions on the device: (iters_done [m], counts [m][k], changed [m], paused);
        comm: row-sharded data set, the all-reduce of every iteration below the C ABI"""
        iters = np.zeros(self.m, np.uint32)
        counts = np.zeros((self.m, self.k), np.uint32)
        changed = np.zeros(self.m, np.uint8)
        paused = C.c_int(0)
        if comm is None:
            check(load().vqhip_kmeans_run(self.raw, int(max_iters), ptr(iters, _u32p), ptr(counts, _u32p), ptr(changed, _u8p),
                                          C.byref(paused)))
        else:
            check(load().vqhip_kmeans_run_sharded(self.raw, comm.raw, int(max_iters), ptr(iters, _u32p), ptr(counts, _u32p),
                                                  ptr(changed, _u8p), C.byref(paused)))
        return iters, counts, changed.astype(bool), bool(paused.value)

    def accumulate(self):
        check(load().vqhip_kmeans_accumulate(self.raw))

    def partials(self):
        """(device pointer, number of f64 elements) of the per-cluster sums/counts slab"""
        p, n = C.c_void_p(), C.c_uint64()
        check(load().vqhip_kmeans_partials(self.raw, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def finalize(self):
        counts = np.empty((self.m, self.k), np.uint32)
        changed = np.empty(self.m, np.uint8)
        check(load().vqhip_kmeans_finalize(self.raw, ptr(counts, _u32p), ptr(changed, _u8p)))
        return counts, changed.astype(bool)

    # -- row-sharded training (RCCL below the ABI) ------------------------------------------
    def allreduce(self, comm: "NativeComm | None"):
        check(load().vqhip_kmeans_allreduce(self.raw, comm.raw if comm is not None else None))

    def step_sharded(self, comm: "NativeComm | None"):
        counts = np.empty((self.m, self.k), np.uint32)
        changed = np.empty(self.m, np.uint8)
        check(load().vqhip_kmeans_step_sharded(self.raw, comm.raw if comm is not None else None,
                                               ptr(counts, _u32p), ptr(changed, _u8p)))
        return counts, changed.astype(bool)

    def init_from_global_rows(self, comm: "NativeComm | None", rows, row_offset: int):
        r = np.ascontiguousarray(rows, dtype=np.uint64).reshape(self.m, self.k)
        check(load().vqhip_kmeans_init_from_global_rows(self.raw, comm.raw if comm is not None else None,
                                                        ptr(r, _u64p), int(row_offset)))

    def patch_from_global_row(self, comm: "NativeComm | None", s: int, j: int, global_row: int, row_offset: int):
        check(load().vqhip_kmeans_patch_from_global_row(self.raw, comm.raw if comm is not None else None,
                                                        int(s), int(j), int(global_row), int(row_offset)))

    def gather_owned_rows(self, rows, row_offset: int) -> np.ndarray:
        """uint32 bit patterns [m][k][sd] of the owned rows among global ids `rows`, zeros elsewhere"""
        r = np.ascontiguousarray(rows, dtype=np.uint64).reshape(self.m, self.k)
        out = np.empty((self.m, self.k, self.sd), np.uint32)
        check(load().vqhip_kmeans_gather_owned_rows(self.raw, ptr(r, _u64p), int(row_offset), ptr(out, _u32p)))
        return out

    def patch_centroid(self, s: int, j: int, sub_row):
        r = f32c(sub_row).reshape(self.sd)
        check(load().vqhip_kmeans_patch_centroid(self.raw, s, j, ptr(r, _f32p)))

    def patch_from_row(self, s: int, j: int, row: int):
        check(load().vqhip_kmeans_patch_from_row(self.raw, s, j, row))

    def get_assignments(self) -> np.ndarray:
        out = np.empty((self.ds.n, self.m), code_dtype(self.k))
        check(load().vqhip_kmeans_get_assignments(self.raw, ptr(out, _u8p)))
        return out


class NativeComm(Handle):
    """vqhip_comm: an RCCL communicator below the C ABI (one rank per GPU).  `uid` is the 128-byte id
    rank 0 obtained from ``NativeComm.unique_id()``; world == 1 with uid None is the identity."""

    _destroy = "vqhip_comm_destroy"
    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * NativeComm.ID_BYTES)()
        check(load().vqhip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes | None, world: int, rank: int):
        h = C.c_void_p()
        if uid is not None:
            if len(uid) != self.ID_BYTES:
                raise FfiError("the RCCL unique id is 128 bytes", ERR_INVALID_INPUT)
            buf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(uid)
            check(load().vqhip_comm_create(buf, int(world), int(rank), C.byref(h)))
        else:
            check(load().vqhip_comm_create(None, int(world), int(rank), C.byref(h)))
        super().__init__(h)
        self.world, self.rank = int(world), int(rank)

    def info(self) -> tuple[int, int]:
        """(world, rank) as the library reports them (read back from RCCL's ncclCommCount / ncclCommUserRank when the
        communicator is a real one)"""
        w, r = C.c_int(0), C.c_int(0)
        check(load().vqhip_comm_info(self.raw, C.byref(w), C.byref(r)))
        return int(w.value), int(r.value)


def _devices(devices) -> np.ndarray:
    """device slots of a one-process multi-GPU handle: None -> every visible device; an int -> that many, 0..n-1"""
    if devices is None:
        devices = range(max(1, device_count()))
    elif isinstance(devices, (int, np.integer)):
        devices = range(int(devices))
    dv = np.ascontiguousarray(list(devices), dtype=np.int32)
    if dv.size == 0:
        raise FfiError("devices is empty", ERR_INVALID_INPUT)
    return dv


class MDataset(Handle):
    """vqhip_mdataset: the rows in contiguous blocks over several devices of THIS process (a worker thread per device
    slot inside the library); `devices` may name a device more than once"""
    _destroy = "vqhip_mdataset_destroy"

    def __init__(self, raw, n: int, d: int, devices: np.ndarray):
        super().__init__(raw)
        self.n, self.d, self.devices = int(n), int(d), devices

    @classmethod
    def from_host(cls, rows: np.ndarray, devices=None) -> "MDataset":
        rows = f32c(rows)
        assert rows.ndim == 2
        dv = _devices(devices)
        raw = C.c_void_p()
        check(load().vqhip_mdataset_from_host(ptr(rows, _f32p), rows.shape[0], rows.shape[1], ptr(dv, _i32p), dv.size, C.byref(raw)))
        return cls(raw, rows.shape[0], rows.shape[1], dv)

    @classmethod
    def synthetic(cls, n: int, d: int, seed: int, devices=None) -> "MDataset":
        dv = _devices(devices)
        raw = C.c_void_p()
        check(load().vqhip_mdataset_synthetic(n, d, seed, ptr(dv, _i32p), dv.size, C.byref(raw)))
        return cls(raw, n, d, dv)

    def rows_per_device(self) -> np.ndarray:
        out = np.zeros(self.devices.size, np.uint64)
        check(load().vqhip_mdataset_info(self.raw, None, None, None, ptr(out, _u64p)))
        return out


class MKMeans(Handle):
    """vqhip_mkmeans: the sharded fit with its ranks inside the library -- KMeans's methods (what pq.fit_codebooks uses),
    row ids GLOBAL"""
    _destroy = "vqhip_mkmeans_destroy"

    def __init__(self, ds: MDataset, m: int, k: int):
        raw = C.c_void_p()
        check(load().vqhip_mkmeans_create(ds.raw, m, k, C.byref(raw)))
        super().__init__(raw)
        self.ds, self.m, self.k, self.sd = ds, m, k, ds.d // m

    def info(self) -> tuple[int, int]:
        """(ranks, communicator kind: 0 identity, 1 RCCL, 2 in-process exchange)"""
        w, kd = C.c_int(0), C.c_int(0)
        check(load().vqhip_mkmeans_info(self.raw, C.byref(w), C.byref(kd)))
        return int(w.value), int(kd.value)

    def set_engine(self, engine: int):
        check(load().vqhip_mkmeans_set_engine(self.raw, engine))

    def set_exact_update(self, on: bool):
        check(load().vqhip_mkmeans_set_exact_update(self.raw, 1 if on else 0))

    def set_centroids(self, c):
        c = f32c(c).reshape(self.m, self.k, self.sd)
        check(load().vqhip_mkmeans_set_centroids(self.raw, ptr(c, _f32p)))

    def init_from_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64).reshape(self.m, self.k)
        check(load().vqhip_mkmeans_init_from_rows(self.raw, ptr(rows, _u64p)))

    def get_centroids(self) -> np.ndarray:
        out = np.empty((self.m, self.k, self.sd), np.float32)
        check(load().vqhip_mkmeans_get_centroids(self.raw, ptr(out, _f32p)))
        return out

    def set_active(self, active):
        a = np.ascontiguousarray(active, dtype=np.uint8).reshape(self.m)
        check(load().vqhip_mkmeans_set_active(self.raw, ptr(a, _u8p)))

    def get_active(self) -> np.ndarray:
        a = np.zeros(self.m, np.uint8)
        check(load().vqhip_mkmeans_get_active(self.raw, ptr(a, _u8p)))
        return a.astype(bool)

    def run(self, max_iters: int):
        iters = np.zeros(self.m, np.uint32)
        counts = np.zeros((self.m, self.k), np.uint32)
        changed = np.zeros(self.m, np.uint8)
        paused = C.c_int(0)
        check(load().vqhip_mkmeans_run(self.raw, int(max_iters), ptr(iters, _u32p), ptr(counts, _u32p), ptr(changed, _u8p), C.byref(paused)))
        return iters, counts, changed.astype(bool), bool(paused.value)

    def patch_from_row(self, s: int, j: int, row: int):
        check(load().vqhip_mkmeans_patch_from_row(self.raw, s, j, row))


class MPQEncoder(Handle):
    """vqhip_mpq_encoder: PQEncoder.encode with the host rows split in row blocks over several devices of this process"""
    _destroy = "vqhip_mpq_encoder_destroy"

    def __init__(self, codebooks, metric: int, devices=None):
        cb = f32c(codebooks)
        assert cb.ndim == 3
        self.m, self.k, self.sd = cb.shape
        self.devices = _devices(devices)
        raw = C.c_void_p()
        check(load().vqhip_mpq_encoder_create(ptr(cb, _f32p), self.m, self.k, self.sd, metric, ptr(self.devices, _i32p),
                                              self.devices.size, C.byref(raw)))
        super().__init__(raw)

    def set_engine(self, engine: int):
        check(load().vqhip_mpq_encoder_set_engine(self.raw, engine))

    def encode(self, rows, want_codes=True, want_f16=True, out_codes=None, out_f16=None):
        rows = f32c(rows).reshape(-1, self.m * self.sd)
        n = rows.shape[0]
        codes, f16 = _encode_outputs(n, self.m, self.k, self.m * self.sd, want_codes, want_f16, out_codes, out_f16)
        check(load().vqhip_mpq_encode(self.raw, ptr(rows, _f32p), n, ptr(codes, _u8p),
                                      None if f16 is None else f16.ctypes.data_as(_u16p)))
        return codes, (None if f16 is None else f16.view(np.float16))

    def decode(self, codes) -> np.ndarray:
        """(n, m) codes -> (n, dim) float32 centroids, row blocks over the devices"""
        codes = np.asarray(codes)
        if codes.size and (codes.min() < 0 or codes.max() >= self.k):
            raise FfiError(f"code outside [0, {self.k})", ERR_INVALID_INPUT)
        codes = np.ascontiguousarray(codes, dtype=code_dtype(self.k)).reshape(-1, self.m)
        out = np.empty((codes.shape[0], self.m * self.sd), np.float32)
        check(load().vqhip_mpq_decode(self.raw, ptr(codes, _u8p), codes.shape[0], ptr(out, _f32p)))
        return out

    def dequantize_f16(self, f16) -> np.ndarray:
        h = np.ascontiguousarray(f16, dtype=np.float16)
        out = np.empty(h.shape, np.float32)
        check(load().vqhip_mpq_dequantize_f16(self.raw, ptr(h.view(np.uint16), _u16p), h.size, ptr(out, _f32p)))
        return out

    def encode_dataset(self, ds: MDataset, repeat: int = 1, want_codes: bool = False):
        """`repeat` passes over the RESIDENT rows of a sharded data set on their devices; the last pass's codes if asked"""
        codes = np.empty((ds.n, self.m), code_dtype(self.k)) if want_codes else None
        check(load().vqhip_mpq_encode_dataset(self.raw, ds.raw, int(repeat), ptr(codes, _u8p)))
        return codes


class MTSVQ(Handle):
    """vqhip_mtsvq: the flattened tree on several devices of this process, host rows descended in row blocks"""
    _destroy = "vqhip_mtsvq_destroy"

    def __init__(self, centroids, left, right, metric: int, devices=None):
        cen = f32c(centroids)
        lf = np.ascontiguousarray(left, dtype=np.int32)
        rt = np.ascontiguousarray(right, dtype=np.int32)
        self.d = int(cen.shape[1])
        self.devices = _devices(devices)
        raw = C.c_void_p()
        check(load().vqhip_mtsvq_create(ptr(cen, _f32p), ptr(lf, _i32p), ptr(rt, _i32p), cen.shape[0], self.d, metric,
                                        ptr(self.devices, _i32p), self.devices.size, C.byref(raw)))
        super().__init__(raw)

    def encode(self, rows, leaf, f16):
        """rows (n, d) f32 C-contiguous; leaf int32 (n,) or None; f16 uint16/float16 (n, d) or None -- filled in place"""
        check(load().vqhip_mtsvq_encode(self.raw, ptr(rows, _f32p), rows.shape[0], ptr(leaf, _i32p),
                                        None if f16 is None else f16.ctypes.data_as(_u16p)))

    def dequantize_f16(self, f16) -> np.ndarray:
        h = np.ascontiguousarray(f16, dtype=np.float16)
        out = np.empty(h.shape, np.float32)
        check(load().vqhip_mtsvq_dequantize_f16(self.raw, ptr(h.view(np.uint16), _u16p), h.size, ptr(out, _f32p)))
        return out

    def last_stats(self):
        scr, und = C.c_int(0), C.c_uint64(0)
        check(load().vqhip_mtsvq_last_stats(self.raw, C.byref(scr), C.byref(und)))
        return bool(scr.value), int(und.value)


def shard_rows(n: int, world: int, rank: int) -> tuple[int, int]:
    """(offset, count) of `rank`'s row block as the library's one-process handles cut it"""
    off, cnt = C.c_uint64(0), C.c_uint64(0)
    check(load().vqhip_shard_rows(n, world, rank, C.byref(off), C.byref(cnt)))
    return int(off.value), int(cnt.value)


class PQEncoder(Handle):
    """vqhip_pq_encoder: batch nearest-centroid encode against fixed codebooks."""

    _destroy = "vqhip_pq_encoder_destroy"

    def __init__(self, codebooks, metric: int):
        cb = f32c(codebooks)
        m, k, sd = cb.shape
        h = C.c_void_p()
        check(load().vqhip_pq_encoder_create(ptr(cb, _f32p), m, k, sd, metric, C.byref(h)))
        super().__init__(h)
        self.m, self.k, self.sd, self.metric = m, k, sd, metric

    def set_engine(self, engine: int):
        check(load().vqhip_pq_encoder_set_engine(self.raw, engine))

    def adc_search(self, codes, queries, topk: int):
        """top-k rows of `codes` [n][m] (u8, u16 above 256 centroids; numpy array, (device_ptr, n), or None = the store adc_set_codes uploaded) per query by asymmetric
        distance; returns (idx uint32 [nq][topk], dist float32 [nq][topk])"""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        idx = np.empty((nq, topk), np.uint32)
        dist = np.empty((nq, topk), np.float32)
        if nq == 0:
            return idx, dist
        lib = load()
        if codes is None:  # the codes adc_set_codes left on the device
            check(lib.vqhip_pq_adc_search_resident(self.raw, ptr(q, _f32p), nq, int(topk), ptr(idx, _u32p), ptr(dist, _f32p)))
        elif isinstance(codes, tuple):
            dev_ptr, n = codes
            check(lib.vqhip_pq_adc_search_device(self.raw, C.c_void_p(dev_ptr), int(n), ptr(q, _f32p), nq, int(topk),
                                                 ptr(idx, _u32p), ptr(dist, _f32p)))
        else:
            c = np.ascontiguousarray(codes, dtype=code_dtype(self.k))
            check(lib.vqhip_pq_adc_search(self.raw, ptr(c, _u8p), c.shape[0], ptr(q, _f32p), nq, int(topk),
                                          ptr(idx, _u32p), ptr(dist, _f32p)))
        return idx, dist

    def adc_set_codes(self, codes) -> None:
        """upload a code store once; adc_search(None, queries, topk) then searches it (codes are checked against k here)"""
        c = np.ascontiguousarray(codes, dtype=code_dtype(self.k))
        check(load().vqhip_pq_adc_set_codes(self.raw, ptr(c, _u8p), c.shape[0]))

    def adc_last_redone(self) -> int:
        """queries of the last adc_search that went through the full pass (all of them where the one-scan schedule does
        not apply)"""
        v = C.c_uint32(0)
        check(load().vqhip_pq_adc_last_redone(self.raw, C.byref(v)))
        return int(v.value)

    def encode(self, rows, want_codes=True, want_f16=True, out_codes=None, out_f16=None):
        """codes (n, m) and / or the f16 reconstruction (n, dim): fresh arrays per call, like the reference's binding
        (pyvq/src/pq.rs:96-107) -- large ones in recycled buffers (vq_amd/_arena.py: the first touch of a new 256 MB array
        was 25 ms of page faults under the copy from the device, against 10 ms for the whole call into pages that exist).
        out_codes / out_f16: C-contiguous arrays of the right shape and dtype to fill instead."""
        rows = f32c(rows).reshape(-1, self.m * self.sd)
        n = rows.shape[0]
        codes, f16 = _encode_outputs(n, self.m, self.k, self.m * self.sd, want_codes, want_f16, out_codes, out_f16)
        check(load().vqhip_pq_encode(self.raw, ptr(rows, _f32p), n, ptr(codes, _u8p),
                                     None if f16 is None else f16.ctypes.data_as(_u16p)))
        return codes, (None if f16 is None else f16.view(np.float16))

    def encode_device(self, dev_rows: int, n: int, dev_codes: int | None, dev_f16: int | None):
        check(load().vqhip_pq_encode_device(self.raw, C.c_void_p(dev_rows), n,
                                            C.c_void_p(dev_codes or 0), C.c_void_p(dev_f16 or 0)))

    def decode(self, codes) -> np.ndarray:
        codes = np.asarray(codes)
        if codes.size and (codes.min() < 0 or codes.max() >= self.k):
            raise FfiError(f"code outside [0, {self.k})", ERR_INVALID_INPUT)
        codes = np.ascontiguousarray(codes, dtype=code_dtype(self.k)).reshape(-1, self.m)
        out = np.empty((codes.shape[0], self.m * self.sd), np.float32)
        check(load().vqhip_pq_decode(self.raw, ptr(codes, _u8p), codes.shape[0], ptr(out, _f32p)))
        return out

    def decode_device(self, dev_codes: int, n: int, dev_out: int):
        """device pointers, asynchronous on the current stream; codes in [0, k) are the caller's responsibility"""
        check(load().vqhip_pq_decode_device(self.raw, C.c_void_p(dev_codes), int(n), C.c_void_p(dev_out)))


def dequantize_f16(f16) -> np.ndarray:
    h = np.ascontiguousarray(f16, dtype=np.float16)
    out = np.empty(h.shape, np.float32)
    check(load().vqhip_dequantize_f16(ptr(h.view(np.uint16), _u16p), h.size, ptr(out, _f32p)))
    return out


def dequantize_f16_device(dev_f16: int, count: int, dev_out: int):
    check(load().vqhip_dequantize_f16_device(C.c_void_p(dev_f16), int(count), C.c_void_p(dev_out)))


def set_device(device: int):
    check(load().vqhip_set_device(device))


def get_device() -> int:
    """the calling thread's current HIP device (0 when no device is visible: the compute calls then report it)"""
    d = C.c_int(0)
    return int(d.value) if load().vqhip_get_device(C.byref(d)) == OK else 0


def set_stream(stream_ptr: int | None):
    check(load().vqhip_set_stream(C.c_void_p(stream_ptr or 0)))


def selftest():
    """(ratio 32x32x16, ratio 16x16x32, trusted): bf16 MFMA accumulation error in units of 2^-24"""
    a, b, t = C.c_float(0), C.c_float(0), C.c_int(0)
    check(load().vqhip_selftest(C.byref(a), C.byref(b), C.byref(t)))
    return float(a.value), float(b.value), bool(t.value)


def mfma_bf16_probe(a_bits, b_bits, c) -> np.ndarray:
    """d[t] of one v_mfma_f32_32x32x16_bf16 per trial: a_bits, b_bits uint16 [t][16], c float32 [t]"""
    a = np.ascontiguousarray(a_bits, np.uint16).reshape(-1, 16)
    b = np.ascontiguousarray(b_bits, np.uint16).reshape(-1, 16)
    cc = np.ascontiguousarray(c, np.float32).reshape(-1)
    assert a.shape == b.shape and a.shape[0] == cc.shape[0]
    d = np.empty(cc.shape[0], np.float32)
    check(load().vqhip_mfma_bf16_probe(ptr(a, _u16p), ptr(b, _u16p), ptr(cc, _f32p), cc.shape[0], ptr(d, _f32p)))
    return d


def mfma_bf16_model(a_bits, b_bits, c) -> np.ndarray:
    """the library's software model of v_mfma_f32_32x32x16_bf16 (host arithmetic): same arguments as the probe"""
    a = np.ascontiguousarray(a_bits, np.uint16).reshape(-1, 16)
    b = np.ascontiguousarray(b_bits, np.uint16).reshape(-1, 16)
    cc = np.ascontiguousarray(c, np.float32).reshape(-1)
    assert a.shape == b.shape and a.shape[0] == cc.shape[0]
    d = np.empty(cc.shape[0], np.float32)
    check(load().vqhip_mfma_bf16_model(ptr(a, _u16p), ptr(b, _u16p), ptr(cc, _f32p), cc.shape[0], ptr(d, _f32p)))
    return d


def mfma_bf16_model_check(trials: int, seed: int = 1):
    """(mismatches, first bad trial) of model == hardware over `trials` device-generated operand sets"""
    bad, first = C.c_uint64(0), C.c_uint64(0)
    check(load().vqhip_mfma_bf16_model_check(int(trials), int(seed), C.byref(bad), C.byref(first)))
    return int(bad.value), int(first.value)


def mfma_bf16_model_failures(trials: int, seed: int, cap: int = 4096):
    """(number of failures, trial ids of the first `cap`) of the device-side model check"""
    ids = np.zeros(cap, np.uint64)
    n = C.c_uint64(0)
    check(load().vqhip_mfma_bf16_model_failures(int(trials), int(seed), ptr(ids, _u64p), cap, C.byref(n)))
    return int(n.value), ids[:min(int(n.value), cap)]


def mfma_bf16_model_case(seed: int, trial: int):
    a, b, c = np.zeros(16, np.uint16), np.zeros(16, np.uint16), np.zeros(1, np.float32)
    check(load().vqhip_mfma_bf16_model_case(int(seed), int(trial), ptr(a, _u16p), ptr(b, _u16p), ptr(c, _f32p)))
    return a, b, c


def synchronize():
    check(load().vqhip_synchronize())


def device_count() -> int:
    return int(load().vqhip_device_count())


def backend() -> str:
    return load().vqhip_backend().decode()


def xfer_lane_calls() -> int:
    """host-batch calls of this process that went through the library's transfer lanes"""
    v = C.c_uint64(0)
    check(load().vqhip_xfer_lane_calls(C.byref(v)))
    return int(v.value)


def set_profiling(on: bool):
    check(load().vqhip_set_profiling(1 if on else 0))


def profile_collect():
    """(calls, primary-stage ms total, re-check ms total) since the last collect"""
    n, a, b = C.c_uint32(), C.c_double(), C.c_double()
    check(load().vqhip_profile_collect(C.byref(n), C.byref(a), C.byref(b)))
    return int(n.value), float(a.value), float(b.value)


def memcpy_device(dst: int, src: int, nbytes: int):
    check(load().vqhip_memcpy_device(C.c_void_p(dst), C.c_void_p(src), nbytes))


def last_assign_stats():
    r, e = C.c_uint64(), C.c_int()
    check(load().vqhip_last_assign_stats(C.byref(r), C.byref(e)))
    return int(r.value), int(e.value)
