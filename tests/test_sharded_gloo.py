"""world_size-2 test of the row-sharded k-means orchestration (vq_amd/sharded.py) on CPU ranks
with the gloo backend: sharding arithmetic, the single all-reduce of the fused f64 slab,
identical convergence decisions on every rank, and the owner-broadcast reseed protocol.

The device back end is replaced by a TEST-ONLY stand-in built from the oracle (tests may use
the oracle as the checker; the product back end is vq_amd.sharded.HipShard = libvqhip).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from vq_amd.sharded import Comm, ShardedKMeans, owner_of, shard_rows  # noqa: E402

F = np.float32


class OracleShard:
    """Stand-in for one rank's device: exact assignment through the oracle, partial sums and
    counts in the same f64 slab layout [m][k][sd+1] that libvqhip produces."""

    def __init__(self, X_local, m, k, row_offset):
        import oracle as O

        self.orc = O.get()
        self.X = np.ascontiguousarray(X_local, F)
        self.m, self.k, self.sd = m, k, X_local.shape[1] // m
        self.n_local, self.row_offset = X_local.shape[0], row_offset
        self.cent = np.zeros((m, k, self.sd), F)
        self.active = np.ones(m, bool)
        self._slab = torch.zeros(m * k * (self.sd + 1), dtype=torch.float64)

    def init_from_values(self, c):
        self.cent = np.array(c, F).reshape(self.m, self.k, self.sd)

    def owned_bits(self, rows):
        out = np.zeros((self.m, self.k, self.sd), np.uint32)
        for s in range(self.m):
            for j in range(self.k):
                out[s, j] = self.owned_sub_row_bits(s, int(rows[s, j]))
        return out

    def owned_sub_row_bits(self, s, global_row):
        r = global_row - self.row_offset
        if not 0 <= r < self.n_local:
            return np.zeros(self.sd, np.uint32)
        return self.X[r, s * self.sd:(s + 1) * self.sd].view(np.uint32).copy()

    def accumulate(self):
        slab = np.zeros((self.m, self.k, self.sd + 1), np.float64)
        for s in range(self.m):
            if not self.active[s]:
                continue
            sub = self.X[:, s * self.sd:(s + 1) * self.sd]
            for i in range(self.n_local):
                j = self.orc.find_nearest(sub[i], self.cent[s])
                slab[s, j, :self.sd] += sub[i]
                slab[s, j, self.sd] += 1
        self._slab.copy_(torch.from_numpy(slab.ravel()))

    def slab(self):
        return self._slab

    def commit_slab(self):
        pass

    def finalize(self):
        slab = self._slab.numpy().reshape(self.m, self.k, self.sd + 1)
        counts = np.zeros((self.m, self.k), np.uint32)
        changed = np.zeros(self.m, bool)
        for s in range(self.m):
            if not self.active[s]:
                continue
            for j in range(self.k):
                c = slab[s, j, self.sd]
                counts[s, j] = int(c)
                if c > 0:
                    new = (slab[s, j, :self.sd] / c).astype(F)
                    if not np.all(np.abs(new - self.cent[s, j]) < F(1e-6)):
                        changed[s] = True
                    self.cent[s, j] = new
        return counts, changed

    def patch_centroid(self, s, j, sub_row):
        self.cent[s, j] = np.asarray(sub_row, F)

    def set_active(self, active):
        self.active = np.array(active, bool)

    def get_centroids(self):
        return self.cent.copy()


def _make_data():
    rng = np.random.default_rng(123)
    X = rng.random((601, 8), dtype=F)  # odd row count: uneven shards
    X[10] = X[3]  # duplicate rows -> a cluster that can never win -> forced reseed
    return X


def _make_data4():
    """world_size 4: 603 rows (shards of 151, 151, 151, 150), three duplicated init rows per
    subspace (several reseeds in one iteration, drawn from different owners) and a -0.0 that
    must survive the integer transport of an init row"""
    rng = np.random.default_rng(321)
    X = rng.random((603, 8), dtype=F)
    X[10] = X[3]
    X[400] = X[3]
    X[599] = X[5]
    X[100, 2] = F(-0.0)
    return X


INIT4 = np.array([[3, 10, 400, 100, 300, 602], [5, 599, 50, 150, 250, 451]])
RESEED4 = [[7, 160, 320, 470, 8, 161, 321, 471, 9, 162], [11, 170, 330, 480, 12, 171, 331, 481, 13, 172]]


def _worker4(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = _make_data4()
        n, m, k = X.shape[0], 2, 6
        off, cnt = shard_rows(n, world, rank)
        shard = OracleShard(X[off:off + cnt], m, k, off)
        skm = ShardedKMeans(shard, n, Comm())
        skm.init_from_global_rows(INIT4)
        np.save(os.path.join(out_dir, f"init_{rank}.npy"), shard.get_centroids())
        cb = skm.fit(7, seed=1, init_rows=INIT4, reseed_rows=RESEED4)
        np.save(os.path.join(out_dir, f"cb_{rank}.npy"), cb)
        np.save(os.path.join(out_dir, f"iters_{rank}.npy"), skm.iters)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = _make_data()
        n, m, k = X.shape[0], 2, 5
        off, cnt = shard_rows(n, world, rank)
        shard = OracleShard(X[off:off + cnt], m, k, off)
        skm = ShardedKMeans(shard, n, Comm())
        init = np.array([[3, 10, 100, 300, 600], [5, 50, 150, 250, 599]])
        reseed = [[7, 8, 9, 11, 12, 13, 14, 15]] * m
        cb = skm.fit(6, seed=1, init_rows=init, reseed_rows=reseed)
        np.save(os.path.join(out_dir, f"cb_{rank}.npy"), cb)
        np.save(os.path.join(out_dir, f"iters_{rank}.npy"), skm.iters)
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    for n in (1, 7, 600, 601, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (o1, c1), (o2, _) in zip(spans, spans[1:]):
                assert o1 + c1 == o2
            for row in (0, n // 2, n - 1):
                r = owner_of(n, world, row)
                assert spans[r][0] <= row < spans[r][0] + spans[r][1]


@pytest.mark.timeout(300)
def test_two_rank_fit_equals_single_process_reference(tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    cb0, cb1 = np.load(tmp_path / "cb_0.npy"), np.load(tmp_path / "cb_1.npy")
    it0, it1 = np.load(tmp_path / "iters_0.npy"), np.load(tmp_path / "iters_1.npy")
    # every rank holds the same codebooks and took the same decisions
    np.testing.assert_array_equal(cb0, cb1)
    np.testing.assert_array_equal(it0, it1)

    # and they equal the un-sharded reference algorithm (oracle) on the full matrix, up to the
    # f64-vs-sequential-f32 summation tolerance
    import oracle as O

    X = _make_data()
    init = np.array([[3, 10, 100, 300, 600], [5, 50, 150, 250, 599]], np.uint64)
    reseed = np.array([[7, 8, 9, 11, 12, 13, 14, 15]] * 2, np.uint64)
    cb_ref, it_ref = O.get().pq_fit(X, 2, 5, 6, init, reseed_rows=reseed)
    assert it0.tolist() == it_ref.tolist()
    assert np.max(np.abs(cb0 - cb_ref)) <= 1e-5


@pytest.mark.timeout(300)
def test_four_rank_uneven_shards_multi_reseed(tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker4, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    cbs = [np.load(tmp_path / f"cb_{r}.npy") for r in range(4)]
    its = [np.load(tmp_path / f"iters_{r}.npy") for r in range(4)]
    inits = [np.load(tmp_path / f"init_{r}.npy") for r in range(4)]
    for r in range(1, 4):
        assert cbs[r].tobytes() == cbs[0].tobytes()
        np.testing.assert_array_equal(its[r], its[0])
        assert inits[r].tobytes() == inits[0].tobytes()
    X = _make_data4()
    # the gathered init rows are the rows' own bits (incl. the sign of -0.0), whoever owns them
    for s in range(2):
        want = X[INIT4[s], s * 4:(s + 1) * 4]
        assert inits[0][s].tobytes() == want.tobytes()
    assert np.signbit(inits[0][0, 3, 2])  # row 100, column 2 = -0.0
    import oracle as O

    cb_ref, it_ref = O.get().pq_fit(X, 2, 6, 7, INIT4.astype(np.uint64), reseed_rows=np.array(RESEED4, np.uint64))
    assert its[0].tolist() == it_ref.tolist()
    assert np.max(np.abs(cbs[0] - cb_ref)) <= 1e-5


def test_single_process_comm_is_noop():
    X = _make_data()
    shard = OracleShard(X, 2, 5, 0)
    skm = ShardedKMeans(shard, X.shape[0], Comm())
    assert skm.comm.world == 1 and not skm.comm.on
    init = np.array([[3, 10, 100, 300, 600], [5, 50, 150, 250, 599]])
    cb = skm.fit(2, seed=1, init_rows=init, reseed_rows=[[7, 8, 9, 11]] * 2)
    assert cb.shape == (2, 5, 4) and np.isfinite(cb).all()
