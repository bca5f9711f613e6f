"""Discovery probes for the accumulation datapath of v_mfma_f32_32x32x16_bf16 (run on the GPU box).

Writes gpurun_out/mfma_probe.npz = operand sets (a, b: bf16 bits [t][16]; c: f32 [t]) and the
hardware's results d [t], in families chosen to separate candidate adder models (alignment point,
kept width, truncation vs rounding, grouping of the 16 products, where C enters).  The model is
then fitted offline (tools/mfma_fit.py) and frozen in tests/mfma_model.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bf16(sign, exp2, mant7):
    """bits of (-1)^sign * 2^exp2 * (1 + mant7/128); exp2 in [-126, 127]"""
    return ((sign.astype(np.uint32) << 15) | ((exp2 + 127).astype(np.uint32) << 7) | (mant7.astype(np.uint32) & 127)).astype(np.uint16)


def f32_from(sign, exp2, mant23):
    bits = (sign.astype(np.uint32) << 31) | ((exp2 + 127).astype(np.uint32) << 23) | (mant23.astype(np.uint32) & 0x7FFFFF)
    return bits.view(np.float32)


def products(rng, t, exp_p, live):
    """a, b [t][16] with product exponents (before mantissa carry) exp_p [t][16]; dead slots are 0 * x"""
    ea = np.floor_divide(exp_p, 2)
    eb = exp_p - ea
    a = bf16(rng.integers(0, 2, (t, 16)), ea, rng.integers(0, 128, (t, 16)))
    b = bf16(rng.integers(0, 2, (t, 16)), eb, rng.integers(0, 128, (t, 16)))
    a = np.where(live, a, np.uint16(0))
    return a, b


def family_sparse(rng, t, nnz, gap_max):
    """nnz non-zero addends among the 17 slots (slot 16 = C), exponents within [-gap_max, 0] of the largest"""
    slots = np.argsort(rng.random((t, 17)), axis=1)[:, :nnz]
    live17 = np.zeros((t, 17), bool)
    np.put_along_axis(live17, slots, True, axis=1)
    e = -rng.integers(0, gap_max + 1, (t, 17))
    first = slots[:, 0]
    e[np.arange(t), first] = 0  # one addend at the top
    a, b = products(rng, t, e[:, :16], live17[:, :16])
    c = f32_from(rng.integers(0, 2, t), e[:, 16], rng.integers(0, 1 << 23, t))
    c = np.where(live17[:, 16], c, np.float32(0))
    return a, b, c.astype(np.float32)


def family_dense(rng, t, window, c_mode):
    e = -rng.integers(0, window + 1, (t, 17))
    live = np.ones((t, 16), bool)
    a, b = products(rng, t, e[:, :16], live)
    if c_mode == "zero":
        c = np.zeros(t, np.float32)
    elif c_mode == "top":  # C dominates: products well below it
        c = f32_from(rng.integers(0, 2, t), np.full(t, 4), rng.integers(0, 1 << 23, t))
    else:
        c = f32_from(rng.integers(0, 2, t), e[:, 16], rng.integers(0, 1 << 23, t))
    return a, b, c.astype(np.float32)


def family_same_sign_small(rng, t, lo, hi):
    """C in [1, 2), all products positive with exponents in [-hi, -lo]: truncation shows as a one-sided error"""
    e = -rng.integers(lo, hi + 1, (t, 16))
    ea = np.floor_divide(e, 2)
    a = bf16(np.zeros((t, 16), np.int64), ea, rng.integers(0, 128, (t, 16)))
    b = bf16(np.zeros((t, 16), np.int64), e - ea, rng.integers(0, 128, (t, 16)))
    c = f32_from(np.zeros(t, np.int64), np.zeros(t, np.int64), rng.integers(0, 1 << 23, t))
    return a, b, c.astype(np.float32)


def family_tiny(rng, t):
    """operands near the bottom of the bf16 / f32 range: subnormal inputs, subnormal products and results"""
    ea = rng.integers(-126, -100, (t, 16))
    eb = rng.integers(-40, 20, (t, 16))
    a = bf16(rng.integers(0, 2, (t, 16)), ea, rng.integers(0, 128, (t, 16)))
    b = bf16(rng.integers(0, 2, (t, 16)), eb, rng.integers(0, 128, (t, 16)))
    sub = rng.random((t, 16)) < 0.2  # bf16 subnormals: exponent field 0, non-zero mantissa
    a = np.where(sub, (a & np.uint16(0x807F)) | np.uint16(1), a)
    live = rng.random((t, 16)) < 0.4
    a = np.where(live, a, np.uint16(0))
    c = f32_from(rng.integers(0, 2, t), rng.integers(-126, -110, t), rng.integers(0, 1 << 23, t))
    c = np.where(rng.random(t) < 0.5, c, np.float32(0)).astype(np.float32)
    csub = (rng.integers(0, 1 << 23, t).astype(np.uint32) | (rng.integers(0, 2, t).astype(np.uint32) << 31)).view(np.float32)
    c = np.where(rng.random(t) < 0.2, csub, c).astype(np.float32)
    return a, b, c


def main():
    from vq_amd import _lib

    _lib.load()
    _lib.set_device(0)
    rng = np.random.default_rng(2026)
    fams = []
    fams.append(("pair40", family_sparse(rng, 120_000, 2, 40)))
    fams.append(("triple30", family_sparse(rng, 80_000, 3, 30)))
    fams.append(("quad12", family_sparse(rng, 40_000, 4, 12)))
    for w in (2, 6, 12, 20, 30, 44):
        fams.append((f"dense{w}", family_dense(rng, 30_000, w, "any")))
    fams.append(("dense8_c0", family_dense(rng, 30_000, 8, "zero")))
    fams.append(("dense20_ctop", family_dense(rng, 30_000, 20, "top")))
    fams.append(("pos_small_8_30", family_same_sign_small(rng, 40_000, 8, 30)))
    fams.append(("pos_small_20_28", family_same_sign_small(rng, 20_000, 20, 28)))
    fams.append(("tiny", family_tiny(rng, 30_000)))
    out = {}
    for name, (a, b, c) in fams:
        d = _lib.mfma_bf16_probe(a, b, c)
        out[name + "_a"], out[name + "_b"], out[name + "_c"], out[name + "_d"] = a, b, c, d
        # quick sanity line: error against the exact sum in units of 2^-24 * (|C| + sum|ab|)
        av = (a.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        bv = (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        with np.errstate(all="ignore"):
            exact = c.astype(np.float64) + (av * bv).sum(axis=1)
            mag = np.abs(c.astype(np.float64)) + np.abs(av * bv).sum(axis=1)
            ratio = np.abs(d.astype(np.float64) - exact) / (2.0 ** -24 * mag + 1e-300)
        print(f"{name:18s} trials {len(c):7d}  worst ratio {np.nanmax(ratio):8.3f}  mean {np.nanmean(ratio):.4f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mfma_probe.npz"), **out)
    print("wrote gpurun_out/mfma_probe.npz")


if __name__ == "__main__":
    main()
