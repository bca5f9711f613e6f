"""include/vq.hpp (the C++ host mirror of the reference's Rust surface) builds with g++, reports
the reference's error texts without a device, and on the GPU produces exactly what the Python
mirror produces for the same seed (both run the same control flow over the same C ABI)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from vq_amd import _lib

    _lib.build_if_missing() if hasattr(_lib, "build_if_missing") else None
    out = tmp_path_factory.mktemp("cpp") / "test_vq_hpp"
    libdir = os.path.join(ROOT, "vq_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_vq_hpp.cpp"), "-o", str(out), "-L", libdir, "-lvqhip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return str(out)


def test_cpp_header_validation_and_rng(exe):
    from vq_amd.rng import HostRng

    r = subprocess.run([exe, "validate"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "VALIDATE_OK" in r.stdout, r.stdout + r.stderr
    rng = HostRng(42)
    want = rng.choose_multiple(1000, 4) + [rng.choose(10)]
    got = [int(x) for x in r.stdout.split("rng ")[1].split("\n")[0].split()]
    assert got == want


@pytest.mark.gpu
def test_cpp_host_equals_python_mirror(exe, tmp_path):
    import vq_amd as pyvq

    n, dim, m, k, iters, seed, depth = 4000, 32, 4, 16, 6, 11, 5
    X = np.random.default_rng(3).random((n, dim), dtype=F)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<7Q", n, dim, m, k, iters, seed, depth))
        f.write(X.tobytes())
    r = subprocess.run([exe, "run", str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RUN_OK" in r.stdout and "gfx950" in r.stdout, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    pq = pyvq.ProductQuantizer(X, m, k, iters, pyvq.Distance.euclidean(), seed)
    np.testing.assert_array_equal(take(F, m * k * (dim // m)).reshape(m, k, dim // m), pq.codebooks)
    np.testing.assert_array_equal(take(np.uint16, n * dim).reshape(n, dim), pq.quantize_batch(X).view(np.uint16))
    np.testing.assert_array_equal(take(np.uint8, n * m).reshape(n, m), pq.encode(X))
    t = pyvq.TSVQ(X, depth, pyvq.Distance.squared_euclidean())
    nodes = int(take(np.uint64, 1)[0])
    cent, left, right = t.tree
    assert nodes == cent.shape[0]
    np.testing.assert_array_equal(take(F, nodes * dim).reshape(nodes, dim), cent)
    np.testing.assert_array_equal(take(np.int32, nodes), left)
    np.testing.assert_array_equal(take(np.int32, nodes), right)
    np.testing.assert_array_equal(take(np.int32, n), t.leaf_ids(X))
    np.testing.assert_array_equal(take(np.uint16, dim), t.quantize(X[1]).view(np.uint16))
    assert off == len(raw)


@pytest.mark.gpu
def test_cpp_const_quantize_from_eight_std_threads(exe, tmp_path, oracle):
    """`const` methods of vq::ProductQuantizer / vq::TSVQ from eight std::threads on ONE object each (the crate's types are
    Send + Sync, src/pq.rs:39-45, src/tsvq.rs:186-191): every per-vector result equals the oracle's for the codebooks /
    tree the program trained"""
    import oracle as O

    n, dim, m, k, iters, seed, depth = 4000, 32, 4, 16, 4, 5, 5
    X = np.random.default_rng(8).random((n, dim), dtype=F)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<7Q", n, dim, m, k, iters, seed, depth))
        f.write(X.tobytes())
    r = subprocess.run([exe, "threads", str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "THREADS_OK" in r.stdout, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    cb = take(F, m * k * (dim // m)).reshape(m, k, dim // m)
    nodes = int(take(np.uint64, 1)[0])
    tree = dict(centroids=take(F, nodes * dim).reshape(nodes, dim), left=take(np.int32, nodes), right=take(np.int32, nodes))
    got_pq = take(np.uint16, n * dim).reshape(n, dim)
    got_tree = take(np.uint16, n * dim).reshape(n, dim)
    assert off == len(raw)
    np.testing.assert_array_equal(got_pq, oracle.pq_encode(O.EUCLIDEAN, X, cb)[1])
    np.testing.assert_array_equal(got_tree, oracle.tsvq_encode(O.SQUARED_EUCLIDEAN, X, tree)[1])


@pytest.mark.gpu
def test_cpp_constructor_over_a_device_list(exe, tmp_path, oracle):
    """vq::ProductQuantizer(rows, ..., devices = {0, 0}): one call, two ranks inside libvqhip (include/vq.hpp); the batch
    calls take the row-block path.  Same iteration structure as the one-device fit, codes / f16 equal to the oracle's for
    the quantizer's own codebooks."""
    import oracle as O

    n, dim, m, k, iters, seed = 140_000, 32, 4, 16, 5, 11
    X = np.random.default_rng(4).random((n, dim), dtype=F)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<7Q", n, dim, m, k, iters, seed, 0))
        f.write(X.tobytes())
    r = subprocess.run([exe, "multi", str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MULTI_OK" in r.stdout, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    ncb = m * k * (dim // m)
    one = np.frombuffer(raw, F, ncb, 0).reshape(m, k, dim // m)
    two = np.frombuffer(raw, F, ncb, ncb * 4).reshape(m, k, dim // m)
    q = np.frombuffer(raw, np.uint16, n * dim, 2 * ncb * 4).reshape(n, dim)
    c = np.frombuffer(raw, np.uint8, n * m, 2 * ncb * 4 + n * dim * 2).reshape(n, m)
    assert np.max(np.abs(one - two)) <= 5e-2  # the same fit up to boundary rows (another grouping of the f64 sums)
    want_c, want_f = oracle.pq_encode(O.EUCLIDEAN, X, two, threads=0)
    assert np.array_equal(c.astype(np.uint32), want_c) and np.array_equal(q, want_f)
