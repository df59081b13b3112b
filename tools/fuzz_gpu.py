#!/usr/bin/env python3
"""Extended fuzz of the MFCC kernels against the oracle on the GPU box (not part of the test suite: ~20 s):
1 048 576 Q15 frames from 1 LSB to clipping, bit for bit; 131 071 frames through variants B, A and TF within the
parity tests' tolerances.   usage (box):  python3 tools/fuzz_gpu.py"""
import sys, numpy as np, time
sys.path.insert(0, '.')
import torch
from edison_amd.context import Context
from edison_amd import _lib
from oracle import oracle
oracle.build()
ctx = Context(0)
rng = np.random.default_rng(2024)
# Q15: 1 M frames, amplitudes 1 LSB .. clipping, DC offsets, bit-exact
n = 1 << 20
bad = 0
for part in range(8):
    m = n // 8
    amp = 10.0 ** rng.uniform(0.0, 5.0, (m, 1))
    x = rng.normal(0.0, 1.0, (m, 1024)) * amp + rng.choice([0.0, 0.0, 300.0, -15000.0, 32767.0], (m, 1))
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(-1)
    got = ctx.mfcc_q15(x)
    ref = oracle.mfcc_q15(x, n_threads=16)
    bad += int((got != ref).any(axis=1).sum())
print("q15 fuzz: %d frames, %d differ" % (n, bad))
# float B and A: 131072 frames (odd count too), tolerance as in the parity tests
# (variant TF -- windowed frames, the two-frame loop's WINDOW instances since round 5 -- with variant A's bar: broadband noise has no band at the float32 transform's rounding floor)
for variant, ov, atol, rtol in ((_lib.MFCC_B, oracle.VARIANT_B, 1e-2, 2e-5), (_lib.MFCC_A, oracle.VARIANT_A, 2e-3, 1e-4), (_lib.MFCC_TF, oracle.VARIANT_TF, 2e-3, 1e-4)):
    m = 131071
    x = np.clip(np.rint(rng.normal(0.0, 1.0, (m, 1024)) * 10.0 ** rng.uniform(0.5, 4.3, (m, 1))), -32768, 32767).astype(np.int16).reshape(-1)
    got = ctx.mfcc(x, variant=variant, n_coef=32)
    ref = oracle.mfcc(x, ov, n_threads=16)
    err = np.abs(got - ref) - (atol + rtol * np.abs(ref))
    print("float variant %d: %d frames, worst excess over tolerance %.3g, max abs err %.3g" % (variant, m, err.max(), np.abs(got - ref).max()))
