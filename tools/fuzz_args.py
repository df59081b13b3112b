#!/usr/bin/env python3
"""Random call shapes of the MFCC entry points against the oracle on the GPU box: frame steps 1..2048 (odd steps take
the unaligned load path), frame counts 1..300 (odd counts end the two-frame kernel on a half pair), coefficient
subsets, variants A / B / B+log / C, KWS with random utterance strides.   usage (box): tools/fuzz_args.py [n [seed]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from edison_amd import _lib  # noqa: E402
from edison_amd.context import Context  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    oracle.build()
    ctx = Context(0)
    model = oracle.Model()
    worst = 0.0
    for case in range(n_cases):
        kind = rng.choice(["A", "B", "Blog", "C", "kws"])
        if kind == "kws":
            n_utt, stride = int(rng.integers(1, 9)), int(rng.integers(31 * 1024, 40000))
            audio = np.clip(rng.normal(0, 10.0 ** rng.uniform(1, 4), (n_utt - 1) * stride + 31 * 1024), -32768, 32767).astype(np.int16)
            r = ctx.kws(audio, n_utt=n_utt, utt_stride=stride)
            o = oracle.cnn(model, r["feat"])
            assert np.array_equal(r["softmax"], o["softmax"]) and np.array_equal(r["argmax"], o["argmax"]), ("kws", n_utt, stride)
            ref = np.stack([oracle.net_input(oracle.mfcc(audio[u * stride:u * stride + 31 * 1024], oracle.VARIANT_B)[:, :13]).reshape(-1) for u in range(n_utt)])
            assert np.abs(ref.astype(int) - r["feat"].astype(int)).max() <= 1, ("kws features", n_utt, stride)
            continue
        n_frames, step = int(rng.integers(1, 301)), int(rng.integers(1, 2049))
        n_coef = int(rng.integers(1, 33))
        x = np.clip(rng.normal(0, 10.0 ** rng.uniform(0.5, 4.3), (n_frames - 1) * step + 1024), -32768, 32767).astype(np.int16)
        if kind == "C":
            got = ctx.mfcc_q15(x, n_frames=n_frames, frame_step=step, n_coef=n_coef)
            ref = oracle.mfcc_q15(x, n_frames=n_frames, frame_step=step)[:, :n_coef]
            assert np.array_equal(got, ref), (kind, n_frames, step, n_coef)
            continue
        variant, ov = (_lib.MFCC_A, oracle.VARIANT_A) if kind == "A" else (_lib.MFCC_B, oracle.VARIANT_B)
        use_log = kind == "Blog"
        got = ctx.mfcc(x, n_frames=n_frames, frame_step=step, variant=variant, n_coef=n_coef, use_log=use_log)
        ref = oracle.mfcc(x, ov, n_frames=n_frames, frame_step=step, use_log=use_log)[:, :n_coef]
        atol, rtol = (2e-3, 1e-4) if (kind == "A" or use_log) else (1e-2, 2e-5)
        excess = float((np.abs(got - ref) - (atol + rtol * np.abs(ref))).max())
        worst = max(worst, excess)
        assert excess <= 0, (kind, n_frames, step, n_coef, excess)
    print("%d random call shapes agree with the oracle (float cases within tolerance, worst margin %.3g; Q15 and CNN bit-exact)" % (n_cases, worst))


if __name__ == "__main__":
    main()
