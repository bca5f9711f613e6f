"""GPU side of the row-sharded driver with a single rank: the libvqhip back end (HipShard), the
zero-copy view of the f64 slab that the all-reduce operates on, and equality of the sharded
control flow with the plain fit.  (world_size 2 is covered on CPU ranks, test_sharded_gloo.py;
the 8-GPU run is the driver's.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


def test_hipshard_slab_aliases_library_buffer_and_fit_matches():
    import torch

    from vq_amd import _lib
    from vq_amd.pq import fit_codebooks
    from vq_amd.sharded import Comm, HipShard, ShardedKMeans

    n, d, m, k = 6000, 64, 4, 32
    X = np.random.default_rng(5).random((n, d), dtype=F)
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.int64)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        _lib.set_stream(stream.cuda_stream)
        try:
            ds = _lib.Dataset.from_host(X)
            shard = HipShard(ds, m, k, 0)
            skm = ShardedKMeans(shard, n, Comm())
            skm.init_from_global_rows(init)
            shard.accumulate()
            t = shard.slab()
            ptr, cnt = shard.km.partials()
            assert t.dtype == torch.float64 and t.numel() == cnt == m * k * (d // m + 1)
            assert shard._slab_alias and t.data_ptr() == ptr  # zero copy
            slab = t.cpu().numpy().reshape(m, k, d // m + 1)
            assert slab[:, :, -1].sum() == n * m  # counts column
            np.testing.assert_allclose(slab[:, :, :-1].sum(axis=1).reshape(-1),
                                       X.astype(np.float64).sum(axis=0), rtol=1e-6)
            shard.commit_slab()
            counts, changed = shard.finalize()
            assert counts.sum() == n * m
            cb = skm.fit(6, seed=3, init_rows=init, reseed_rows=[[1] * 32] * m)
            cb_ref = fit_codebooks(ds, m, k, 6, init_rows=init.astype(np.uint64), reseed_rows=[[1] * 32] * m)
            np.testing.assert_array_equal(cb, cb_ref)
            shard.close()
            ds.close()
        finally:
            _lib.set_stream(None)


_RCCL_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VQ_REPO"])
import torch
import torch.distributed as dist
from vq_amd import _lib
from vq_amd.pq import fit_codebooks
from vq_amd.sharded import Comm, HipShard, ShardedKMeans

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
_lib.load(); _lib.set_device(0)
n, d, m, k = 20000, 64, 4, 32
X = np.random.default_rng(6).random((n, d), dtype=np.float32)
init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.int64)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    _lib.set_stream(stream.cuda_stream)
    ds = _lib.Dataset.from_host(X)
    shard = HipShard(ds, m, k, 0)
    skm = ShardedKMeans(shard, n, Comm(force=True))       # 1-rank group, collectives ON
    assert skm.comm.on and skm._collective_device() is not None
    cb = skm.fit(5, seed=3, init_rows=init, reseed_rows=[[7] * 32] * m)
    assert shard._slab_alias
    cb_ref = fit_codebooks(ds, m, k, 5, init_rows=init.astype(np.uint64), reseed_rows=[[7] * 32] * m)
    np.testing.assert_array_equal(cb, cb_ref)
    torch.cuda.synchronize()
    _lib.set_stream(None)
dist.destroy_process_group()
print("RCCL_OK")
'''


_RCCL_WORKER_NO_STREAM = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VQ_REPO"])
import torch
import torch.distributed as dist
from vq_amd import _lib
from vq_amd.pq import fit_codebooks
from vq_amd.sharded import Comm, HipShard, NativeShardedKMeans, ShardedKMeans, native_comm_from_torch

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
_lib.load(); _lib.set_device(0)
n, d, m, k = 200000, 64, 4, 32
X = np.random.default_rng(6).random((n, d), dtype=np.float32)
X[17] = X[4]; X[4000] = X[4]
init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.int64)
init[0, 1], init[0, 2], init[0, 0] = 17, 4000, 4          # duplicates: reseeds in iteration 1
ds = _lib.Dataset.from_host(X)
# (a) the caller never touches vqhip_set_stream: HipShard itself orders the library against torch's stream
shard = HipShard(ds, m, k, 0)
skm = ShardedKMeans(shard, n, Comm(force=True))
cb = skm.fit(5, seed=3, init_rows=init, reseed_rows=[[7, 9, 11, 13] * 8] * m)
_lib.set_stream(None)
cb_ref = fit_codebooks(ds, m, k, 5, init_rows=init.astype(np.uint64), reseed_rows=[[7, 9, 11, 13] * 8] * m)
np.testing.assert_array_equal(cb, cb_ref)
# (b) the collective below the C ABI: RCCL communicator created by the library from a unique id
comm = native_comm_from_torch(force=True)
assert comm.world == 1 and comm.raw
nk = NativeShardedKMeans(ds, m, k, n, 0, comm)
cb2 = nk.fit(5, seed=3, init_rows=init, reseed_rows=[[7, 9, 11, 13] * 8] * m)
np.testing.assert_array_equal(cb2, cb_ref)
assert nk.iters.tolist() == skm.iters.tolist()
nk.close(); comm.close()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_native_comm_in_process_single_rank():
    """vqhip_comm_unique_id / _create / _kmeans_step_sharded / _init_from_global_rows /
    _patch_from_global_row with a real one-rank RCCL communicator, no torch.distributed at all:
    the fit equals the plain single-GPU fit bit for bit (a one-rank sum is the identity)."""
    from vq_amd import _lib
    from vq_amd.pq import fit_codebooks
    from vq_amd.sharded import NativeShardedKMeans

    n, d, m, k = 50_000, 64, 4, 32
    X = np.random.default_rng(8).random((n, d), dtype=F)
    X[33] = X[2]
    X[9, 37] = F(-0.0)  # column 5 of subspace 2
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.int64)
    init[1, 0], init[1, 1], init[2, 3] = 2, 33, 9
    ds = _lib.Dataset.from_host(X)
    cb_ref = fit_codebooks(ds, m, k, 4, init_rows=init.astype(np.uint64), reseed_rows=[[5, 6, 7, 8] * 4] * m)
    for uid in (None, _lib.NativeComm.unique_id()):
        comm = _lib.NativeComm(uid, 1, 0)
        nk = NativeShardedKMeans(ds, m, k, n, 0, comm)
        nk.init_from_global_rows(init)
        c0 = nk.km.get_centroids()
        assert c0[2, 3].tobytes() == X[9, 32:48].tobytes() and np.signbit(c0[2, 3, 5])
        cb = nk.fit(4, seed=1, init_rows=init, reseed_rows=[[5, 6, 7, 8] * 4] * m)
        np.testing.assert_array_equal(cb, cb_ref)
        bits = nk.km.gather_owned_rows(np.where(init < n // 2, init, init + n), 0)  # rows >= n are someone else's
        want = np.where((init < n // 2)[:, :, None],
                        np.stack([X[init[s], s * 16:(s + 1) * 16] for s in range(m)]).view(np.uint32), 0)
        np.testing.assert_array_equal(bits, want)
        nk.close()
        comm.close()
    ds.close()


@pytest.mark.parametrize("which", ["user_stream", "no_stream_and_native"])
def test_collectives_over_rccl_single_rank(tmp_path, which):
    """The all-reduce / broadcast calls of the sharded fit on a real RCCL process group (one
    rank, collectives forced on): same stream, zero-copy slab, identical codebooks."""
    import os
    import subprocess
    import sys

    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER if which == "user_stream" else _RCCL_WORKER_NO_STREAM)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29531" if which == "user_stream" else "29532", HSA_ENABLE_IPC_MODE_LEGACY="0",
               VQ_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
