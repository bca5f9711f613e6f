"""CPU-side checks of the drop-in boundary: libvqhip.so builds for gfx950, loads, exports every
symbol include/vqhip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from vq_amd import _lib

    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "vqhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqhip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} declared in vqhip.h but not exported"
        assert name in built.SIGNATURES, f"{name} has no ctypes signature in vq_amd/_lib.py"
    assert set(built.SIGNATURES) <= set(names)


def test_backend_string_and_error_channel(built):
    assert "gfx950" in built.backend()
    assert isinstance(built.last_error(), str)


def test_code_object_is_gfx950_only(built):
    blob = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"gfx1100", b"sm_90"):
        assert other not in blob


def _gpu_present():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="this check is about machines WITHOUT a GPU")
def test_compute_fails_loudly_without_gpu(built):
    from vq_amd import FfiError, ProductQuantizer

    assert built.device_count() == 0
    with pytest.raises(FfiError) as e:
        built.Dataset.from_host(np.zeros((4, 4), np.float32))
    assert e.value.status == built.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(FfiError):
        built.PQEncoder(np.zeros((2, 2, 2), np.float32), built.EUCLIDEAN)
    with pytest.raises(FfiError):
        ProductQuantizer(np.random.rand(20, 8).astype(np.float32), 2, 4)


def test_null_and_shape_checks_do_not_need_a_gpu(built):
    lib = built.load()
    assert lib.vqhip_dataset_from_host(None, 4, 4, None) == built.ERR_NULL_PTR
    h = ctypes.c_void_p()
    assert lib.vqhip_dataset_from_host(None, 0, 4, ctypes.byref(h)) == built.ERR_INVALID_INPUT
    assert "empty" in built.last_error()
    assert lib.vqhip_kmeans_create(None, 2, 2, ctypes.byref(h)) == built.ERR_NULL_PTR


def test_tsvq_create_rejects_arrays_that_are_not_one_tree(built):
    """shared children, two parents, unreachable nodes and backward edges are refused before any device
    work (the encoder's breadth-first images are sized by the node count)."""
    lib = built.load()
    cent = np.zeros((4, 2), np.float32)
    i32p = ctypes.POINTER(ctypes.c_int32)
    f32p = ctypes.POINTER(ctypes.c_float)

    def create(left, right, metric=0):
        l, r = np.array(left, np.int32), np.array(right, np.int32)
        h = ctypes.c_void_p()
        rc = lib.vqhip_tsvq_create(cent.ctypes.data_as(f32p), l.ctypes.data_as(i32p), r.ctypes.data_as(i32p), len(l), 2, metric,
                                   ctypes.byref(h))
        return rc, built.last_error()

    for left, right, word in (([1, -1, -1, -1], [1, -1, -1, -1], "both children"),    # one node as both children
                              ([1, 3, 3, -1], [2, -1, -1, -1], "two parents"),         # a shared grandchild
                              ([1, -1, -1, -1], [2, -1, -1, -1], "not reachable"),     # node 3 hangs loose
                              ([1, 0, -1, -1], [2, 3, -1, -1], "out-of-order"),        # an edge back to the root
                              ([1, -1, -1, -1], [9, -1, -1, -1], "out-of-order")):     # past the end
        rc, msg = create(left, right)
        assert rc == built.ERR_INVALID_INPUT and word in msg, (left, right, rc, msg)
    rc, msg = create([1, -1, -1, -1], [2, 3, -1, -1], metric=7)
    assert rc == built.ERR_INVALID_INPUT and "metric" in msg


def test_synthetic_generator_host_twin(built):
    a = built.synth_uniform_host(100, 16, seed=66, row_offset=0)
    b = built.synth_uniform_host(40, 16, seed=66, row_offset=60)
    np.testing.assert_array_equal(a[60:], b)  # depends only on (seed, global row, col)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert len(np.unique(a)) > 1500
    # multiples of 2^-24 exactly
    assert np.all(a * 2 ** 24 == np.floor(a * 2 ** 24))


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "vq_oracle" not in text and "import oracle" not in text, f


def test_shard_rows_matches_the_python_drivers():
    """the row blocks of the one-process multi-GPU handles (vqhip_shard_rows) are those of vq_amd/sharded.py"""
    from vq_amd import _lib
    from vq_amd.sharded import shard_rows

    for n in (1, 7, 8, 1000, 1_000_003, 100_000_000):
        for world in (1, 2, 3, 4, 8, 16):
            covered = 0
            for rank in range(world):
                got = _lib.shard_rows(n, world, rank)
                assert got == tuple(shard_rows(n, world, rank)), (n, world, rank)
                assert got[0] == covered
                covered += got[1]
            assert covered == n
