"""GPU parity of the TSVQ build and descent against the oracle: the tree (structure AND
centroid bits) and the leaves must be identical -- the build keeps the reference's sequential
f32 column sums, so even near-tied split dimensions resolve the same way."""
import os

import numpy as np
import pytest

import oracle as O
from vq_amd import TSVQ, Distance, _lib
from vq_amd.tsvq import build_tree

pytestmark = pytest.mark.gpu
F = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _data(seed, n, d, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d), dtype=F)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(F)
    if kind == "lattice":
        return rng.integers(0, 4, (n, d)).astype(F)
    if kind == "structured":  # the reference's own test data, src/tsvq.rs:287-289
        return np.array([[(i + j) % 50 for j in range(d)] for i in range(n)], F)
    raise ValueError(kind)


def _assert_same_tree(got, want):
    cent, left, right = got
    np.testing.assert_array_equal(left, want["left"])
    np.testing.assert_array_equal(right, want["right"])
    a, b = cent, want["centroids"]
    assert a.shape == b.shape
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{(~same).sum()} centroid components differ"


@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice", "structured"])
@pytest.mark.parametrize("shape", [(1000, 32, 5), (3001, 12, 8), (700, 128, 3), (50, 6, 4), (5000, 20, 0), (2, 4, 2), (1, 4, 3)])
def test_build_bit_identical(oracle, kind, shape):
    n, d, depth = shape
    X = _data(21, n, d, kind)
    ds = _lib.Dataset.from_host(X)
    got = build_tree(ds, depth)
    ds.close()
    _assert_same_tree(got, oracle.tsvq_build(X, depth))


@pytest.mark.parametrize("kind", ["normal", "normal_scales", "lattice0", "uniform0"])
@pytest.mark.parametrize("shape", [(70_001, 40, 4), (150_003, 36, 5), (33_000, 128, 2)])
def test_build_zero_mean_long_nodes_bit_identical(oracle, kind, shape):
    """Zero-mean columns on nodes long enough for the exact parallel column sums (>= 16384 rows): the running sum is a
    random walk that keeps re-crossing binade edges and meets exact ties in practically every segment, so the chain runs
    its two-stream path, scans failing lanes over eight lanes and re-adds parked segments from LDS -- on row counts that
    are no multiple of 64, column counts that are no multiple of 32, values over twelve binades, a centred lattice (exact
    sums, ties everywhere) and centred uniform rows.  Structure and every centroid bit against the oracle."""
    n, d, depth = shape
    rng = np.random.default_rng(n + d)
    if kind == "normal":
        X = rng.standard_normal((n, d)).astype(F)
    elif kind == "normal_scales":
        X = (rng.standard_normal((n, d)) * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(F)
    elif kind == "lattice0":
        X = rng.integers(-2, 3, (n, d)).astype(F)
    else:
        X = (rng.random((n, d), dtype=F) - F(0.5)).astype(F)
    ds = _lib.Dataset.from_host(X)
    got = build_tree(ds, depth)
    ds.close()
    _assert_same_tree(got, oracle.tsvq_build(X, depth))


@pytest.mark.parametrize("shape", [(150_003, 96, 4), (90_001, 160, 3)])
def test_build_mixed_columns_both_chain_forms_and_the_policy_cache(oracle, shape):
    """Column blocks of both kinds in one data set -- zero-mean blocks (exact guess: tables + items + the one-pass chain,
    round 5) next to blocks with |mean| >= sigma (sampled guess: round 4's chain) -- so that both forms run in the same
    mean pass, each on its own columns; a last block mixes both kinds of column (one zero-mean column makes the block an
    exact one).  Built three times on ONE library-owned data set: the second and third builds take the sampling policy
    from the data set's cache (no k_fs_policy, only the forms that have columns are launched) and must give the first
    one's bits; a borrowed device buffer (no cache) gives them too."""
    import torch

    n, d, depth = shape
    rng = np.random.default_rng(n + d)
    X = np.empty((n, d), F)
    for c0 in range(0, d, 32):
        blk = (c0 // 32) % 3
        w = min(32, d - c0)
        if blk == 0:
            X[:, c0:c0 + w] = rng.standard_normal((n, w)).astype(F)
        elif blk == 1:
            X[:, c0:c0 + w] = rng.random((n, w), dtype=F) + F(0.25)
        else:
            X[:, c0:c0 + w] = rng.random((n, w), dtype=F) * F(3.0)
            X[:, c0] = rng.standard_normal(n).astype(F)
    want = oracle.tsvq_build(X, depth)
    ds = _lib.Dataset.from_host(X)
    first = build_tree(ds, depth)
    _assert_same_tree(first, want)
    for _ in range(2):
        again = build_tree(ds, depth)
        for a, b in zip(first, again):
            assert a.tobytes() == b.tobytes()
    ds.close()
    Xd = torch.from_numpy(X).cuda()
    borrowed = _lib.Dataset.from_device(Xd.data_ptr(), n, d, keepalive=Xd)
    _assert_same_tree(build_tree(borrowed, depth), want)
    Xd.mul_(-1.0)  # the caller changes the rows behind a borrowed handle: nothing about them is kept between builds
    torch.cuda.synchronize()
    _assert_same_tree(build_tree(borrowed, depth), oracle.tsvq_build(-X, depth))
    borrowed.close()


def test_build_uneven_split_repeats_with_both_paths(oracle, capfd, monkeypatch):
    """The build launches, per level, only the column-sum path that evenly split nodes would take, and checks the level
    table afterwards.  70 % of the rows tie on the widest column: the root splits 140k / 60k, two levels on long and
    short nodes share a level -- the build must notice, repeat itself with both paths (VQHIP_TSVQ_VERBOSE says so), give
    the oracle's tree, and a second build on the same data set must go straight to the conservative form."""
    n, d, depth = 200_000, 16, 5
    rng = np.random.default_rng(77)
    X = rng.random((n, d), dtype=F)
    X[:, 3] = np.where(rng.random(n) < 0.7, F(0.0), F(40.0))
    want = oracle.tsvq_build(X, depth)
    sizes = []
    monkeypatch.setenv("VQHIP_TSVQ_VERBOSE", "1")
    ds = _lib.Dataset.from_host(X)
    got = build_tree(ds, depth)
    first = capfd.readouterr().err
    again = build_tree(ds, depth)
    second = capfd.readouterr().err
    ds.close()
    _assert_same_tree(got, want)
    _assert_same_tree(again, want)
    assert "repeated with both column-sum paths" in first
    assert "repeated" not in second


def test_build_partial_nan_and_identical_rows(oracle):
    X = _data(22, 500, 8, "normal")
    X[17, 3] = np.nan
    X[200:260] = X[200]
    ds = _lib.Dataset.from_host(X)
    got = build_tree(ds, 6)
    ds.close()
    _assert_same_tree(got, oracle.tsvq_build(X, 6))
    # identical vectors: root is a leaf (src/tsvq.rs:273-284)
    Y = np.tile(np.array([1, 2, 3, 4, 5], F), (10, 1))
    t = TSVQ(Y, 3, Distance.squared_euclidean())
    assert t.tree[0].shape[0] == 1
    assert np.all(np.abs(t.quantize(Y[0]).astype(F) - Y[0]) < 1e-2)


def test_build_all_nan_split_column_reports_reference_panic():
    X = np.full((3, 1), np.nan, F)
    with pytest.raises(_lib.FfiError) as e:
        ds = _lib.Dataset.from_host(X)
        build_tree(ds, 2)
    assert "panics" in str(e.value)


@pytest.mark.parametrize("metric", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice"])
def test_descent_bit_identical(oracle, metric, kind):
    X = _data(23, 4000, 24, kind)
    Q = _data(24, 3000, 24, kind)
    tree = oracle.tsvq_build(X, 7)
    names = {0: "squared_euclidean", 1: "euclidean", 2: "manhattan", 3: "cosine"}
    t = TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], Distance(names[metric]))
    want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, tree, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)
    np.testing.assert_array_equal(t.quantize(Q[5]).view(np.uint16), want_f16[5])


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice", "structured"])
@pytest.mark.parametrize("shape", [(6000, 32, 9), (5000, 64, 7), (6000, 128, 8), (3000, 256, 5), (400, 128, 12),
                                   (3000, 384, 5), (2500, 192, 6), (2000, 512, 4), (2000, 768, 5),  # 384: the reference's eval
                                   (5000, 100, 7), (4000, 300, 5), (3000, 20, 6), (3000, 200, 6), (2000, 4, 5),  # zero-padded widths
                                   (1500, 1024, 4), (1200, 1000, 5)])  # 1024 wide
def test_screened_descent_bit_identical(oracle, metric, kind, shape):
    """Squared-L2 / Euclidean descent with d in {32,64,128,192,256,384,512,768}: one dot product per level decides
    the rows whose margin is provable, the rest resume from their node in exact arithmetic."""
    n, d, depth = shape
    X = _data(25, n, d, kind)
    Q = np.concatenate([_data(26, 5000, d, kind), X[:500]])  # training rows sit on cell boundaries more often
    tree = oracle.tsvq_build(X, depth)
    names = {0: "squared_euclidean", 1: "euclidean"}
    t = TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], Distance(names[metric]))
    want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, tree, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    screened, undecided = t.last_encode_stats()
    if tree["centroids"].shape[0] > 1 and (tree["left"] >= 0).sum() * d * 4 < 140 * 1024:
        assert screened
        if kind in ("uniform", "normal"):
            assert undecided < 0.25 * len(Q)
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)


@pytest.mark.parametrize("shape", [(30000, 128, 11), (9000, 768, 7), (20000, 64, 13)])
@pytest.mark.parametrize("kind", ["uniform", "normal"])
def test_screened_descent_beyond_the_lds_levels(oracle, shape, kind):
    """A tree too large for one CU's LDS: the levels nearest the root are screened from LDS, the deeper ones from
    L2; leaves and f16 outputs stay the oracle's."""
    n, d, depth = shape
    X = _data(27, n, d, kind)
    Q = np.concatenate([_data(28, 4000, d, kind), X[:300]])
    tree = oracle.tsvq_build(X, depth)
    n_int = int(((tree["left"] >= 0) & (tree["right"] >= 0)).sum())
    assert n_int * d * 4 > 170 * 1024  # really larger than the LDS
    t = TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], Distance("squared_euclidean"))
    want_leaf, want_f16 = oracle.tsvq_encode(0, Q, tree, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    screened, undecided = t.last_encode_stats()
    assert screened and undecided < 0.5 * len(Q)
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)


def test_screened_descent_adversarial(oracle):
    """Queries on the bisecting planes (exact ties -> left), at the centroids, non-finite, huge
    and denormal magnitudes; a tree with a non-finite centroid."""
    rng = np.random.default_rng(27)
    n, d, depth = 4000, 64, 6
    X = rng.standard_normal((n, d)).astype(F)
    tree = oracle.tsvq_build(X, depth)
    cent, left, right = tree["centroids"], tree["left"], tree["right"]
    inner = np.where((left >= 0) & (right >= 0))[0]
    mid = ((cent[left[inner]].astype(np.float64) + cent[right[inner]]) / 2).astype(F)  # on the plane, up to rounding
    Q = np.concatenate([mid, cent, mid + F(1e-7), mid - F(1e-7), rng.standard_normal((2000, d)).astype(F) * F(1e18),
                        rng.standard_normal((500, d)).astype(F) * F(1e-30),
                        rng.standard_normal((500, d)).astype(F) * F(1e-41), np.zeros((3, d), F)])
    Q[7, 5] = np.nan
    Q[8, 9] = np.inf
    Q[9, 1] = -np.inf
    for metric, name in ((0, "squared_euclidean"), (1, "euclidean")):
        t = TSVQ.from_tree(cent, left, right, Distance(name))
        want_leaf, _ = oracle.tsvq_encode(metric, Q, tree, want_f16=False, threads=0)
        np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
        assert t.last_encode_stats()[0]
    bad = {k: v.copy() for k, v in tree.items()}
    bad["centroids"][3, 2] = np.nan
    bad["centroids"][5, 0] = np.inf
    t = TSVQ.from_tree(bad["centroids"], left, right, Distance("squared_euclidean"))
    want_leaf, _ = oracle.tsvq_encode(0, Q, bad, want_f16=False, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)


@pytest.mark.parametrize("metric", [0, 3])
def test_folded_continuation_with_full_lists(oracle, metric):
    """Round 6: a wave finishes its own undecided rows.  Here most rows are undecided -- on or within an ulp of the root's
    bisecting plane -- so every wave's list (64 entries in LDS) fills up several times and the tile loop is left and
    re-entered; leaves and f16 rows equal the oracle's, and the two-kernel form's (VQHIP_TSVQ_FOLD is read once per
    process: compared through the oracle)."""
    rng = np.random.default_rng(31)
    n0, d, depth = 6000, 128, 8
    X = rng.standard_normal((n0, d)).astype(F)
    tree = oracle.tsvq_build(X, depth)
    cent, left, right = tree["centroids"], tree["left"], tree["right"]
    cl, cr = cent[left[0]].astype(np.float64), cent[right[0]].astype(np.float64)
    # squared L2: the midpoint of the root's children; cosine: the bisector of their directions (equal angles to both)
    mid = ((cl + cr) / 2).astype(F) if metric == 0 else (cl / np.linalg.norm(cl) + cr / np.linalg.norm(cr)).astype(F)
    n = 600_000  # 4096 waves x 146 rows: lists of 64 overflow in every wave
    Q = np.repeat(mid[None, :], n, axis=0)
    Q[::3] += (rng.standard_normal((len(Q[::3]), d)) * 1e-7).astype(F)   # a third a hair off the plane
    Q[1::50] = rng.standard_normal((len(Q[1::50]), d)).astype(F)          # and some ordinary rows in between
    name = {0: "squared_euclidean", 3: "cosine"}[metric]
    t = TSVQ.from_tree(cent, left, right, Distance(name))
    want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, tree, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    screened, undecided = t.last_encode_stats()
    assert screened and undecided > 0.5 * n  # the lists really were full
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)


@pytest.mark.parametrize("kind", ["uniform", "normal", "lattice", "structured"])
@pytest.mark.parametrize("shape", [(6000, 32, 9), (6000, 128, 8), (3000, 384, 5), (2500, 192, 6), (2000, 768, 5),
                                   (5000, 100, 7), (3000, 20, 6), (2000, 4, 5), (1500, 1024, 4), (30000, 128, 11)])
@pytest.mark.parametrize("metric", [3, 2])
def test_screened_cosine_manhattan_descent_bit_identical(oracle, kind, shape, metric):
    """Cosine descent: two dot products with the children's unit vectors decide the rows whose margin is provable
    (q_l >= q_r, or q_r < 0 where the clamp makes d_r = 1; strictly right only when 1 - q survives its rounding), the
    rest resume in the reference's arithmetic (three sequential sums, EPSILON rule, clamp).  Manhattan descent: both
    L1 sums in a tree order, decided when they differ by more than the two orders' relative error bounds."""
    n, d, depth = shape
    X = _data(35, n, d, kind)
    Q = np.concatenate([_data(36, 5000, d, kind), X[:500], -_data(37, 300, d, kind)])
    tree = oracle.tsvq_build(X, depth)
    t = TSVQ.from_tree(tree["centroids"], tree["left"], tree["right"], Distance({3: "cosine", 2: "manhattan"}[metric]))
    want_leaf, want_f16 = oracle.tsvq_encode(metric, Q, tree, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    screened, undecided = t.last_encode_stats()
    assert screened
    if kind in ("uniform", "normal") and d >= 20:
        assert undecided < 0.25 * len(Q)
    np.testing.assert_array_equal(t.quantize_batch(Q).view(np.uint16), want_f16)


def test_screened_manhattan_descent_adversarial(oracle):
    """Exact L1 ties (-> left), rows at centroids (distance 0), sums that overflow, denormal and non-finite rows,
    a tree with non-finite centroids."""
    rng = np.random.default_rng(39)
    n, d, depth = 4000, 64, 6
    X = rng.integers(-3, 4, (n, d)).astype(F)  # small integers: every L1 sum is exact, ties are real
    tree = oracle.tsvq_build(X, depth)
    cent, left, right = tree["centroids"], tree["left"], tree["right"]
    inner = np.where((left >= 0) & (right >= 0))[0]
    mid = ((cent[left[inner]].astype(np.float64) + cent[right[inner]]) / 2).astype(F)
    Q = np.concatenate([mid, cent, X[:1500], rng.integers(-3, 4, (1500, d)).astype(F),
                        rng.standard_normal((300, d)).astype(F) * F(1e37), rng.standard_normal((300, d)).astype(F) * F(3e38),
                        rng.standard_normal((300, d)).astype(F) * F(1e-38), rng.standard_normal((300, d)).astype(F) * F(1e-42),
                        np.zeros((3, d), F), rng.standard_normal((1000, d)).astype(F)])
    Q[7, 5] = np.nan
    Q[8, 9] = np.inf
    Q[9, 1] = -np.inf
    t = TSVQ.from_tree(cent, left, right, Distance("manhattan"))
    want_leaf, _ = oracle.tsvq_encode(2, Q, tree, want_f16=False, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    assert t.last_encode_stats()[0]
    bad = {k: v.copy() for k, v in tree.items()}
    bad["centroids"][3, 2] = np.nan
    bad["centroids"][5, 0] = np.inf
    bad["centroids"][int(left[inner[1]])] *= F(1e38)
    t = TSVQ.from_tree(bad["centroids"], left, right, Distance("manhattan"))
    want_leaf, _ = oracle.tsvq_encode(2, Q, bad, want_f16=False, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)


def test_screened_cosine_descent_adversarial(oracle):
    """Rows parallel / anti-parallel / orthogonal to centroids (q = 1, q < 0 -> clamp ties -> left), equal cosines,
    norms around the reference's EPSILON = 1e-10 and around the overflow of the squared norm, zero, denormal and
    non-finite rows; trees with a zero, a tiny, a huge and a non-finite centroid."""
    rng = np.random.default_rng(38)
    n, d, depth = 4000, 64, 6
    X = rng.standard_normal((n, d)).astype(F)
    tree = oracle.tsvq_build(X, depth)
    cent, left, right = tree["centroids"], tree["left"], tree["right"]
    inner = np.where((left >= 0) & (right >= 0))[0]
    cl, cr = cent[left[inner]].astype(np.float64), cent[right[inner]].astype(np.float64)
    ul, ur = cl / np.linalg.norm(cl, axis=1, keepdims=True), cr / np.linalg.norm(cr, axis=1, keepdims=True)
    bis = (ul + ur).astype(F)  # equal cosines up to rounding
    Q = np.concatenate([bis, bis * F(3.5), -bis, cent, cent * F(-2), cent * F(1e-3), (ul - ur).astype(F),
                        bis + F(1e-7), bis - F(1e-7),
                        rng.standard_normal((300, d)).astype(F) * F(1e-11), rng.standard_normal((300, d)).astype(F) * F(2e-12),
                        rng.standard_normal((300, d)).astype(F) * F(1.2e-11 / 8), rng.standard_normal((300, d)).astype(F) * F(1e-30),
                        rng.standard_normal((300, d)).astype(F) * F(1e-41), rng.standard_normal((300, d)).astype(F) * F(1e18),
                        rng.standard_normal((300, d)).astype(F) * F(2.3e18), rng.standard_normal((300, d)).astype(F) * F(1e30),
                        np.zeros((3, d), F), rng.standard_normal((2000, d)).astype(F)])
    Q[7, 5] = np.nan
    Q[8, 9] = np.inf
    Q[9, 1] = -np.inf
    t = TSVQ.from_tree(cent, left, right, Distance("cosine"))
    want_leaf, _ = oracle.tsvq_encode(3, Q, tree, want_f16=False, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    assert t.last_encode_stats()[0]
    for edit in ("zero", "tiny", "huge", "nan", "inf", "negated"):
        bad = {k: v.copy() for k, v in tree.items()}
        c = bad["centroids"]
        node = int(left[inner[1]])
        if edit == "zero":
            c[node] = 0
        elif edit == "tiny":
            c[node] *= F(1e-12)
        elif edit == "huge":
            c[node] *= F(1e19)
        elif edit == "nan":
            c[node, 2] = np.nan
        elif edit == "inf":
            c[node, 0] = np.inf
        else:
            c[int(right[inner[0]])] *= F(-1)
        t = TSVQ.from_tree(c, left, right, Distance("cosine"))
        want_leaf, _ = oracle.tsvq_encode(3, Q, bad, want_f16=False, threads=0)
        np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf, err_msg=edit)


@pytest.mark.parametrize("metric", ["cosine", "manhattan"])
def test_cosine_manhattan_descent_fullsize_eval_shape(metric):
    """The reference's `make eval ALG=tsvq` shape under cosine / Manhattan (1M x 384, depth 5): every leaf of the
    screened descent equals the all-exact walk of the same library (which the smaller cases pin to the oracle)."""
    n, d, depth = 1_000_000, 384, 5
    ds = _lib.Dataset.synthetic(n, d, seed=67)
    cent, left, right = build_tree(ds, depth)
    X = ds.read()
    ds.close()
    t = TSVQ.from_tree(cent, left, right, Distance(metric))
    got = t.leaf_ids(X)
    screened, undecided = t.last_encode_stats()
    assert screened and undecided < 0.2 * n
    os.environ["VQHIP_TSVQ_EXACT"] = "1"
    try:
        t_exact = TSVQ.from_tree(cent, left, right, Distance(metric))
    finally:
        del os.environ["VQHIP_TSVQ_EXACT"]
    np.testing.assert_array_equal(got, t_exact.leaf_ids(X))
    assert not t_exact.last_encode_stats()[0]


def test_golden_tsvq_fixture():
    g = np.load(os.path.join(GOLD, "tsvq_depth5.npz"))
    ds = _lib.Dataset.from_host(g["X"])
    cent, left, right = build_tree(ds, 5)
    ds.close()
    assert cent.tobytes() == g["centroids"].tobytes()
    np.testing.assert_array_equal(left, g["left"])
    np.testing.assert_array_equal(right, g["right"])
    for metric, mname, dn in ((0, "sqeuclid", "squared_euclidean"), (1, "euclid", "euclidean"),
                              (2, "manhattan", "manhattan"), (3, "cosine", "cosine")):
        t = TSVQ.from_tree(cent, left, right, Distance(dn))
        np.testing.assert_array_equal(t.leaf_ids(g["Q"]), g[f"leaf_{mname}"])
        np.testing.assert_array_equal(t.quantize_batch(g["Q"]).view(np.uint16), g[f"f16_{mname}"])


def test_config4_fullsize_depth8(oracle):
    """BASELINE config 4: TSVQ depth 8 on 1M x 128.  The oracle needs ~10 s for the build;
    compare the whole tree, then the descent of a sample."""
    n, d, depth = 1_000_000, 128, 8
    ds = _lib.Dataset.synthetic(n, d, seed=66)
    cent, left, right = build_tree(ds, depth)
    X = ds.read()
    ds.close()
    want = oracle.tsvq_build(X, depth)
    _assert_same_tree((cent, left, right), want)
    assert cent.shape[0] == 511
    t = TSVQ.from_tree(cent, left, right, Distance.euclidean())
    Q = X[::997]
    want_leaf, _ = oracle.tsvq_encode(O.EUCLIDEAN, Q, want, want_f16=False, threads=0)
    np.testing.assert_array_equal(t.leaf_ids(Q), want_leaf)
    screened, undecided = t.last_encode_stats()
    assert screened and undecided < 0.1 * len(Q)
    # every row: the screened descent against the all-exact walk of the same library
    all_leaf = t.leaf_ids(X)
    os.environ["VQHIP_TSVQ_EXACT"] = "1"
    try:
        t_exact = TSVQ.from_tree(cent, left, right, Distance.euclidean())
    finally:
        del os.environ["VQHIP_TSVQ_EXACT"]
    np.testing.assert_array_equal(all_leaf, t_exact.leaf_ids(X))
    assert not t_exact.last_encode_stats()[0]
