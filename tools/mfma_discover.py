"""Discovery probes for the accumulation datapath of v_mfma_f32_32x32x16_bf16 (run on the GPU box).

Writes gpurun_out/mfma_probe.npz = operand sets (a, b: bf16 bits [t][16]; c: f32 [t]) and the
hardware's results d [t] for the families of tests/mfma_families.py.  The adder model was fitted to
this data offline (tools/mfma_fit.py) and is frozen in vq_amd/csrc/mfma_model.hpp (C++, used by the
library's self-test) and tests/mfma_model.py (independent Python statement).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from mfma_families import all_families
    from vq_amd import _lib

    _lib.load()
    _lib.set_device(0)
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
    out = {}
    for name, (a, b, c) in all_families(rng):
        d = _lib.mfma_bf16_probe(a, b, c)
        m = _lib.mfma_bf16_model(a, b, c)
        out[name + "_a"], out[name + "_b"], out[name + "_c"], out[name + "_d"] = a, b, c, d
        av = (a.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        bv = (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        with np.errstate(all="ignore"):
            exact = c.astype(np.float64) + (av * bv).sum(axis=1)
            mag = np.abs(c.astype(np.float64)) + np.abs(av * bv).sum(axis=1)
            ratio = np.abs(d.astype(np.float64) - exact) / (2.0 ** -24 * mag + 1e-300)
            ratio = ratio[np.isfinite(ratio) & np.isfinite(d)]
        same = (m.view(np.uint32) == d.view(np.uint32)) | ((m == 0) & (d == 0))
        print(f"{name:18s} trials {len(c):7d}  model mismatches {int((~same).sum()):6d}  "
              f"worst |d-exact|/(2^-24 mag) {ratio.max() if ratio.size else 0:8.3f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mfma_probe2.npz"), **out)
    print("wrote gpurun_out/mfma_probe2.npz")
    bad, first = _lib.mfma_bf16_model_check(1 << 30, 7)
    print(f"device check: 2^30 generated operand sets, {bad} mismatches (first bad trial {first})")


if __name__ == "__main__":
    main()
