"""vq_amd/_arena.py: the recycled buffers behind the batch calls' fresh result arrays (the reference's binding returns a
new array per call, pyvq/src/pq.rs:96-107).  A buffer may go back to the pool only when NO array refers to it."""
import gc

import numpy as np

from vq_amd import _arena as A


def test_small_results_are_plain_arrays():
    a = A.fresh((100, 16), np.float16)
    assert a.base is None and a.shape == (100, 16) and a.dtype == np.float16


def test_buffer_returns_only_when_every_view_is_gone():
    A.clear()
    shape = (1 << 19, 16)  # 16 MB of f16
    a = A.fresh(shape, np.float16)
    assert a.shape == shape and a.dtype == np.float16 and a.flags.c_contiguous and a.flags.writeable
    a[:] = 2.5
    where = a.ctypes.data
    piece, bits = a[7:9, 3:5], a.view(np.uint16)  # numpy collapses their .base to the array that owns the lease
    del a, bits
    gc.collect()
    assert A._idle_bytes == 0                      # the slice still refers to the memory: not recycled
    b = A.fresh(shape, np.float16)
    assert b.ctypes.data != where                  # a second result gets other memory
    b[:] = -1.0
    assert float(piece[0, 0]) == 2.5
    del piece
    gc.collect()
    assert A._idle_bytes > 0
    c = A.fresh(shape, np.float16)
    assert c.ctypes.data == where                  # same size: the first buffer again, pages intact
    del b, c
    gc.collect()
    A.clear()
    assert A._idle_bytes == 0


def test_pool_is_bounded():
    A.clear()
    keep = [A.fresh((1 << 19, 16), np.float16) for _ in range(A._PER_SIZE + 3)]
    del keep
    gc.collect()
    assert A._idle_bytes <= A._PER_SIZE * (16 << 20) and A._idle_bytes <= A._CAP_BYTES
    A.clear()
