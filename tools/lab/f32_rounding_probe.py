#!/usr/bin/env python3
"""Variant D against the reference-made fixture (tests/golden/mfccf32_golden.npz): where the int8 outputs differ, how far is the GPU's own
pre-rounding value (coefficient x 2^dec_bits) from a rounding boundary? (round-3 review item 7: 97 % exact for configuration 1 against
99.5 % for the others -- a stage with an extra rounding distance, or the float32 noise floor?)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from edison_amd.context import Context
from edison_amd.mfcc.mfcc_f32 import MfccF32
g = np.load(os.path.join(ROOT, "tests", "golden", "mfccf32_golden.npz"))
ctx = Context(0)
for i in range(3):
    nf, off, flen, bits, pre, hop = g["cfg%d" % i]
    m = MfccF32(ctx=ctx, num_mfcc_features=int(nf), feature_offset=int(off), frame_len=int(flen), mfcc_dec_bits=int(bits), preemph=float(pre))
    got, f32, lm = m.compute(g["audio"], frame_step=int(hop), want_float=True)
    ref = g["mfcc%d" % i]
    x = f32.astype(np.float64)                      # what the kernel rounds (already scaled by 2^dec_bits?)
    scaled = x if np.abs(x).max() > 4 else x * 2.0 ** bits
    mis = got != ref
    unsat = (np.abs(ref.astype(int)) < 127) & (np.abs(got.astype(int)) < 127)
    dist = np.abs(np.abs(scaled - np.trunc(scaled)) - 0.5)     # distance of the pre-rounding value from x.5
    print("cfg%d frame %d dec_bits %d offset %d preemph %.2f: %d values, %.3f %% differ (all by 1: %s); unsaturated values %.1f %%" % (
        i, flen, bits, off, pre, ref.size, 100 * mis.mean(), bool((np.abs(got.astype(int) - ref.astype(int))[mis] == 1).all()), 100 * unsat.mean()))
    if mis.any():
        print("   distance of the GPU's pre-rounding value from a rounding boundary, differing values: max %.5f, 99th pct %.5f, median %.5f LSB" % (
            dist[mis].max(), np.percentile(dist[mis], 99), np.median(dist[mis])))
        for thr in (0.001, 0.003, 0.01, 0.03):
            near = (dist < thr) & unsat
            print("   values within %.3f LSB of a boundary: %.3f %% of the unsaturated ones; of those %.1f %% differ" % (thr, 100 * near.sum() / max(unsat.sum(), 1), 100 * (mis & near).sum() / max(near.sum(), 1)))
        # per coefficient
        print("   differing fraction per coefficient: " + " ".join("%.2f" % (100 * mis[:, c].mean()) for c in range(ref.shape[1])))
        print("   |scaled value| median per coefficient:  " + " ".join("%.0f" % np.median(np.abs(scaled[:, c])) for c in range(ref.shape[1])))
    if i == 1:
        # the values that differ far from a boundary: which frames are they?
        far = mis & (dist > 0.01)
        fr = np.flatnonzero(far.any(axis=1))
        print("   cfg1: %d frames hold the %d values that differ more than 0.01 LSB from a boundary; frame indices %s ..." % (len(fr), far.sum(), fr[:12]))
        rl = g["logmel1"]
        print("   their reference log-mel: min over bands, median over those frames %.2f (all frames: %.2f); max |sample| of those frames: %s" % (
            np.median(rl[fr].min(axis=1)), np.median(rl.min(axis=1)),
            sorted(set(int(np.abs(g["audio"][f * int(hop): f * int(hop) + int(flen)].astype(int)).max()) for f in fr))[:8]))
        print("   max |log-mel GPU - reference| over those frames %.3g, over the others %.3g" % (np.abs(lm[fr] - rl[fr]).max(), np.abs(np.delete(lm, fr, 0) - np.delete(rl, fr, 0)).max()))
    m.close()
