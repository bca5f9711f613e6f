"""Full-size GPU checks at BASELINE.json's configurations through size-independent properties, plus an oracle
spot check on a random sample of rows.  (Every row against the oracle: tests/test_gpu_allrows.py -- the GPU box's 128
host threads scan 1M x 128 x 256 in about a second.)

  * engine agreement: MFMA screen + exact re-check == exact scan, bit for bit, on every row;
  * idempotence / k = N identity: encoding the codebook rows returns each row's own index;
  * checksum of checksums: per-cluster counts of a Lloyd step sum to N in every subspace and
    equal the histogram of the assignment codes;
  * mean property: sum_j counts_j * centroid_j == column sums of X (within f32 tolerance);
  * the encode of a row does not depend on which batch it travels in (ragged re-batching).
"""
import numpy as np
import pytest

import oracle as O
from vq_amd import _lib

pytestmark = pytest.mark.gpu
F = np.float32

CONFIGS = {
    "C2": dict(n=1_000_000, d=128, m=8, k=256, metric=_lib.SQUARED_EUCLIDEAN),
    "C3_cosine_reduced_n": dict(n=200_000, d=768, m=96, k=256, metric=_lib.COSINE),
    "C5_per_gpu_reduced_n": dict(n=1_000_000, d=128, m=16, k=256, metric=_lib.SQUARED_EUCLIDEAN),
}


def _trained(ds, m, k, iters=2):
    n = ds.n
    km = _lib.KMeans(ds, m, k)
    init = np.array([[(j * (n // k) + s) % n for j in range(k)] for s in range(m)], np.uint64)
    km.init_from_rows(init)
    for _ in range(iters):
        counts, changed = km.step()
    return km, counts


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_fullsize_encode_properties(oracle, cfg):
    c = CONFIGS[cfg]
    n, d, m, k, metric = c["n"], c["d"], c["m"], c["k"], c["metric"]
    sd = d // m
    ds = _lib.Dataset.synthetic(n, d, seed=66)
    km, counts = _trained(ds, m, k)
    cb = km.get_centroids()

    # Lloyd-step bookkeeping: counts == histogram of codes, sum to N
    assign = km.get_assignments()
    for s in range(m):
        np.testing.assert_array_equal(np.bincount(assign[:, s], minlength=k), counts[s])
    assert (counts.sum(axis=1) == n).all()
    km.close()

    import torch

    enc = _lib.PQEncoder(cb, metric)
    codes_a = torch.empty((n, m), dtype=torch.uint8, device="cuda")
    codes_b = torch.empty((n, m), dtype=torch.uint8, device="cuda")
    enc.set_engine(_lib.ENGINE_AUTO)
    enc.encode_device(ds.device_ptr, n, codes_a.data_ptr(), None)
    _lib.synchronize()
    rechecked, engine = _lib.last_assign_stats()
    enc.set_engine(_lib.ENGINE_EXACT)
    enc.encode_device(ds.device_ptr, n, codes_b.data_ptr(), None)
    _lib.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(codes_a, codes_b)  # engines agree on every row
    if metric in (_lib.SQUARED_EUCLIDEAN, _lib.EUCLIDEAN):
        assert engine in (_lib.ENGINE_MFMA, _lib.ENGINE_MFMA_BF16) and rechecked < 0.05 * n * m

    # oracle spot check on a random sample of rows
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(n, 1500, replace=False))
    Xs = np.stack([ds.read(int(r), 1)[0] for r in rows[:300]])
    want, _ = oracle.pq_encode(metric, Xs, cb, want_f16=False, threads=0)
    got = codes_a.cpu().numpy()[rows[:300]]
    np.testing.assert_array_equal(got.astype(np.uint32), want)

    # batching independence: the same rows encoded as a small host batch
    got2, f16 = enc.encode(Xs)
    np.testing.assert_array_equal(got2, got)
    np.testing.assert_array_equal(f16, np.concatenate(
        [cb[s][got[:, s]] for s in range(m)], axis=1).astype(np.float16))

    # idempotence: each centroid encodes to itself (first index among exact duplicates)
    cents = np.concatenate([cb[s] for s in range(m)], axis=1)  # row j = centroid j of every subspace
    self_codes, _ = enc.encode(cents, want_f16=False)
    want_self, _ = oracle.pq_encode(metric, cents, cb, want_f16=False, threads=0)
    np.testing.assert_array_equal(self_codes.astype(np.uint32), want_self)
    enc.close()
    ds.close()


def test_fullsize_mean_property():
    n, d, m, k = 1_000_000, 128, 8, 256
    sd = d // m
    ds = _lib.Dataset.synthetic(n, d, seed=66)
    km, counts = _trained(ds, m, k, iters=1)
    cent = km.get_centroids()
    X = ds.read()
    col = X.astype(np.float64).sum(axis=0)
    for s in range(m):
        rec = (counts[s][:, None].astype(np.float64) * cent[s].astype(np.float64)).sum(axis=0)
        np.testing.assert_allclose(rec, col[s * sd:(s + 1) * sd], rtol=2e-6)
    km.close()
    ds.close()


@pytest.mark.parametrize("kind", ["clustered", "heavy_tail", "offset", "tiny_scale"])
@pytest.mark.parametrize("shape", [(400_000, 128, 8, 256), (300_000, 384, 16, 256), (300_000, 96, 8, 100),
                                   (300_000, 128, 4, 256), (200_000, 192, 4, 256), (200_000, 128, 2, 256)])
def test_fullsize_engine_agreement_distributions(kind, shape):
    """Screen + re-check == exact scan on every row for data that stresses the margin: tight
    clusters (genuine near-ties), heavy tails (a few huge norms), a common offset far from the
    origin, and values near the f32 underflow range.  Squared-L2 and cosine; sub_dim 16 / 24 / 12."""
    import torch

    n, d, m, k = shape
    g = torch.Generator(device="cuda").manual_seed(97)
    if kind == "clustered":
        centers = torch.randn((k, d), device="cuda", generator=g)
        X = centers[torch.randint(0, k, (n,), device="cuda", generator=g)] + \
            0.01 * torch.randn((n, d), device="cuda", generator=g)
    elif kind == "heavy_tail":
        X = torch.randn((n, d), device="cuda", generator=g) * torch.exp(2.5 * torch.randn((n, 1), device="cuda", generator=g))
    elif kind == "offset":
        X = torch.rand((n, d), device="cuda", generator=g) + 517.25
    else:
        X = torch.rand((n, d), device="cuda", generator=g) * 1e-19
    X = X.to(torch.float32).contiguous()
    torch.cuda.synchronize()
    ds = _lib.Dataset.from_device(X.data_ptr(), n, d, keepalive=X)
    km, _ = _trained(ds, m, k, iters=2)
    cb = km.get_centroids()
    km.close()
    for metric in (_lib.SQUARED_EUCLIDEAN, _lib.COSINE):
        enc = _lib.PQEncoder(cb, metric)
        a = torch.empty((n, m), dtype=torch.uint8, device="cuda")
        b = torch.empty((n, m), dtype=torch.uint8, device="cuda")
        enc.set_engine(_lib.ENGINE_AUTO)
        enc.encode_device(ds.device_ptr, n, a.data_ptr(), None)
        _lib.synchronize()
        _, engine = _lib.last_assign_stats()
        assert engine == _lib.ENGINE_MFMA_BF16
        enc.set_engine(_lib.ENGINE_EXACT)
        enc.encode_device(ds.device_ptr, n, b.data_ptr(), None)
        _lib.synchronize()
        torch.cuda.synchronize()
        diff = int((a != b).sum().item())
        assert diff == 0, f"{diff} codes differ ({kind}, metric {metric})"
        enc.close()
    ds.close()
