"""Dump the operand sets on which the device-side model check (vqhip_mfma_bf16_model_check) disagrees with the
hardware: operands, hardware result and model result -> gpurun_out/mfma_failures.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vq_amd import _lib  # noqa: E402

_lib.load()
_lib.set_device(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
n, ids = _lib.mfma_bf16_model_failures(trials, seed)
print(f"seed {seed}: {n} failures of {trials}")
A, B, Cc = [], [], []
for t in ids:
    a, b, c = _lib.mfma_bf16_model_case(seed, int(t))
    A.append(a), B.append(b), Cc.append(c[0])
if ids.size:
    A, B, Cc = np.array(A), np.array(B), np.array(Cc, np.float32)
    hw = _lib.mfma_bf16_probe(A, B, Cc)
    md = _lib.mfma_bf16_model(A, B, Cc)
    print("families of the failures:", np.bincount((ids & 7).astype(np.int64), minlength=8))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mfma_failures.npz"), ids=ids, a=A, b=B, c=Cc, hw=hw, model=md)
