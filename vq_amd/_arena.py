"""Recycled host buffers behind the batch calls' fresh result arrays.

The reference's binding returns a NEW numpy array from every call (pyvq/src/pq.rs:96-107), and so does this package;
but the first touch of a fresh 256 MB array -- page faults and zeroing under the copy from the device -- costs more
than the whole encode (25 of 35 ms at 1M x 128).  Large results therefore live in buffers that come back to a small
pool when the LAST array referring to them dies and are handed out again, pages intact, to the next call of the same
size; a call that finds the pool empty allocates as before (asking the kernel for the pages ahead of the copies --
madvise(MADV_POPULATE_WRITE) from the transfer lanes, or from a team of 8 / 16 extra threads -- was measured at no gain
or a loss: profiles/NOTES.md, round 5).

Safety: numpy collapses the `.base` of a view of a view to the object that owns the memory.  That owner is a `_Lease`
made per hand-out (not the pooled buffer, not another array), so every array derived from a result -- slices, reshapes,
`.view()`s -- keeps the lease alive, and the buffer goes back only when none of them is left."""
from __future__ import annotations

import os
import threading
import weakref

import numpy as np

_MIN_BYTES = 8 << 20                                              # smaller results: plain np.empty
_CAP_BYTES = int(os.environ.get("VQ_AMD_ARENA_MB", "1024")) << 20  # pooled (idle) bytes kept at most; 0 disables the pool
_PER_SIZE = 4
_GRAIN = 2 << 20

# re-entrant: _give_back runs from weakref.finalize, i.e. possibly inside a garbage collection that an allocation under the
# lock triggered on the same thread; the critical sections below also allocate no GC-tracked object
_lock = threading.RLock()
_idle: dict[int, list[np.ndarray]] = {}
_idle_bytes = 0
stats = {"reused": 0, "allocated": 0, "dropped": 0}


class _Lease:
    """owner of one hand-out: exposes the pooled buffer through the array interface, returns it when collected"""
    __slots__ = ("_buf", "__array_interface__", "__weakref__")

    def __init__(self, buf: np.ndarray, shape, dtype):
        self._buf = buf
        self.__array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": np.dtype(dtype).str,
                                    "data": (buf.ctypes.data, False), "version": 3}


def _give_back(buf: np.ndarray) -> None:
    global _idle_bytes
    spare: list = []  # (made outside the lock: a list is GC-tracked, its allocation may start a collection)
    with _lock:
        lst = _idle.get(buf.nbytes)
        if lst is None:
            lst = _idle[buf.nbytes] = spare
        if len(lst) < _PER_SIZE and _idle_bytes + buf.nbytes <= _CAP_BYTES:
            lst.append(buf)
            _idle_bytes += buf.nbytes
        else:
            stats["dropped"] += 1


def fresh(shape, dtype) -> np.ndarray:
    """a C-contiguous array of `shape` / `dtype` with undefined contents (np.empty's contract)"""
    global _idle_bytes
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < _MIN_BYTES or _CAP_BYTES <= 0:
        return np.empty(shape, dtype)
    size = (nbytes + _GRAIN - 1) // _GRAIN * _GRAIN
    with _lock:
        lst = _idle.get(size)
        buf = lst.pop() if lst else None
        if buf is not None:
            _idle_bytes -= size
            stats["reused"] += 1
        else:
            stats["allocated"] += 1
    if buf is None:
        buf = np.empty(size, np.uint8)
    lease = _Lease(buf, shape, dtype)
    weakref.finalize(lease, _give_back, buf)
    return np.asarray(lease)


def clear() -> None:
    """drop the idle buffers (tests; memory pressure)"""
    global _idle_bytes
    with _lock:
        _idle.clear()
        _idle_bytes = 0
