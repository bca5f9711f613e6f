"""Quantizer::dequantize for batches (src/pq.rs:201-209): codes -> f32 centroids (vqhip_pq_decode[_device]) and f16 -> f32
(vqhip_dequantize_f16[_device]) against numpy, over the kernels' vector and scalar forms (sub_dim % 4, two-byte codes,
element counts around the 8-wide groups, unaligned device pointers) and the row-block forms over device slots."""
import numpy as np
import pytest

from vq_amd import _lib
from vq_amd.errors import FfiError

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("m,k,sd", [(8, 256, 16), (3, 17, 6), (4, 300, 4), (1, 2, 1), (96, 256, 8), (2, 5, 12)])
@pytest.mark.parametrize("n", [1, 1000, 70001])
def test_decode_equals_the_gather(m, k, sd, n):
    rng = np.random.default_rng(m * 1000 + k + n)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    cb[0, 0, 0] = np.float32(-0.0)
    cb[m - 1, k - 1, sd - 1] = np.float32(1e-42)  # subnormal: copied, not flushed
    codes = rng.integers(0, k, (n, m)).astype(_lib.code_dtype(k))
    codes[0, :] = k - 1
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    got = enc.decode(codes)
    want = np.concatenate([cb[s][codes[:, s]] for s in range(m)], axis=1)
    assert got.shape == (n, m * sd)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    with pytest.raises(FfiError):
        bad = codes.astype(np.int64)
        bad[n // 2, 0] = k
        enc.decode(bad)
    enc.close()


@pytest.mark.parametrize("m,k,sd,n", [(4, 300, 4, 300_001), (2, 5, 12, 200_003), (8, 256, 16, 100_003), (16, 64, 8, 40_961), (9, 100, 16, 33_000)])
def test_decode_with_the_codebooks_in_lds(m, k, sd, n):
    """the form with the codebooks in LDS (at least 2^20 16-byte groups, codebooks up to 144 KB): two-byte codes, a last trip
    that ends inside a row, m not a power of two"""
    rng = np.random.default_rng(n)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(_lib.code_dtype(k))
    codes[-1, :] = k - 1
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    got = enc.decode(codes)
    want = np.concatenate([cb[s][codes[:, s]] for s in range(m)], axis=1)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    enc.close()


def test_decode_device_form_and_unaligned_output():
    import torch

    m, k, sd, n = 8, 256, 16, 50_000
    rng = np.random.default_rng(3)
    cb = rng.standard_normal((m, k, sd)).astype(F)
    codes = rng.integers(0, k, (n, m)).astype(np.uint8)
    want = np.concatenate([cb[s][codes[:, s]] for s in range(m)], axis=1)
    enc = _lib.PQEncoder(cb, _lib.SQUARED_EUCLIDEAN)
    dc = torch.from_numpy(codes).cuda()
    torch.cuda.synchronize()
    buf = torch.zeros(n * m * sd + 1, dtype=torch.float32, device="cuda")
    for off in (0, 1):  # 16-byte aligned: one float4 per lane; 4 bytes off: the scalar form
        out = buf[off:off + n * m * sd]
        out.zero_()
        torch.cuda.synchronize()  # (the library launches on its own stream: torch's fill must have landed)
        enc.decode_device(dc.data_ptr(), n, out.data_ptr())
        _lib.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().reshape(n, m * sd).view(np.uint32), want.view(np.uint32))
    enc.close()


@pytest.mark.parametrize("count", [1, 7, 8, 9, 15, 16, 17, 4099, 1_000_003])
def test_dequantize_f16_all_counts(count):
    rng = np.random.default_rng(count)
    h = rng.standard_normal(count).astype(np.float16)
    special = np.array([np.inf, -np.inf, np.nan, 6e-8, -6e-8, 65504.0, -0.0, 0.0], np.float16)
    h[:min(count, special.size)] = special[:min(count, special.size)]
    got = _lib.dequantize_f16(h)
    np.testing.assert_array_equal(got.view(np.uint32), h.astype(F).view(np.uint32))  # exact, NaN payloads included


def test_dequantize_f16_device_form_unaligned():
    import torch

    count = 100_003
    h = np.random.default_rng(9).standard_normal(count + 8).astype(np.float16)
    dh = torch.from_numpy(h.view(np.int16)).cuda()
    out = torch.empty(count + 8, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for src_off, dst_off in ((0, 0), (1, 0), (0, 1), (3, 2)):  # element offsets: 2-byte / 4-byte misalignment of the 16-byte form
        src = dh[src_off:src_off + count]
        dst = out[dst_off:dst_off + count]
        dst.zero_()
        torch.cuda.synchronize()  # (the library launches on its own stream: torch's fill must have landed)
        _lib.dequantize_f16_device(src.data_ptr(), count, dst.data_ptr())
        _lib.synchronize()
        np.testing.assert_array_equal(dst.cpu().numpy().view(np.uint32), h[src_off:src_off + count].astype(F).view(np.uint32))
