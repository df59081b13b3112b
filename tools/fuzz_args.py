#!/usr/bin/env python3
"""Random call shapes of the MFCC entry points against the oracle on the GPU box: frame steps 1..2048 (odd steps take
the unaligned load path), frame counts 1..300 (odd counts end the two-frame kernel on a half pair), coefficient
subsets, variants A / B / B+log / C, KWS with random utterance strides; since round 5 also random LISTS of batches in one launch
(edison_mfcc_batches_dev: 1..40 batches, odd frame counts, one batch 2-byte aligned only) and the two-queue calls against one call per
batch, bit for bit, and the generality kernel (edison_mfcc_generic: random frame lengths 4..1500, steps, mel bins, edges, scales) against
oracle.mfcc_numpy (numpy's FFT like the reference; pinned on the reference's own outputs for six geometries).   usage (box): tools/fuzz_args.py [n [seed]]"""
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
        kind = rng.choice(["A", "B", "Blog", "C", "kws", "list", "queues", "generic"])
        if kind == "generic":
            N, nm = int(rng.integers(4, 1501)), int(rng.integers(1, 81))
            step, n = int(rng.integers(1, 2 * N + 1)), int(rng.integers(1, 12))
            fs = float(rng.choice([8000.0, 16000.0, 44100.0]))
            lo = float(rng.uniform(0.0, 0.1 * fs)); hi = float(rng.uniform(lo + 0.05 * fs, 0.5 * fs)); scale = float(rng.choice([1.0, 16.0, 128.0, 1000.0]))
            variant = [_lib.MFCC_A, _lib.MFCC_B, _lib.MFCC_TF][int(rng.integers(0, 3))]     # TF: unpinned, held to the same numpy restatement
            use_log = bool(variant == _lib.MFCC_B and rng.random() < 0.5)
            x = np.clip(rng.normal(0, 10.0 ** rng.uniform(0.5, 4.3), (n - 1) * step + N), -32768, 32767).astype(np.int16)
            got = np.zeros((n, nm))
            r_ = _lib.lib().edison_mfcc_generic(ctx._h, x.ctypes.data, n, N, step, variant | (_lib.MFCC_USE_LOG if use_log else 0), nm, fs, lo, hi, scale,
                                                None, None, None, None, got.ctypes.data, 0, None, 1.0)
            assert r_ == 0, (kind, N, nm, r_)
            ref = oracle.mfcc_numpy(x, {_lib.MFCC_A: oracle.VARIANT_A, _lib.MFCC_B: oracle.VARIANT_B, _lib.MFCC_TF: oracle.VARIANT_TF}[variant], N, step, n_frames=n, num_mel_bins=nm, sample_rate=fs,
                                    lower_edge_hertz=lo, upper_edge_hertz=hi, mel_mtx_scale=scale, use_log=use_log)   # pinned on the reference's outputs: tests/test_oracle.py
            assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), (kind, N, step, n, nm, variant, use_log, np.abs(got - ref).max())
            continue
        if kind in ("list", "queues"):
            nb, n_each, step = int(rng.integers(1, 41 if kind == "list" else 7)), int(rng.integers(1, 120)), int(rng.choice([1024, 512, 1000, 333]))
            variant, use_log = (_lib.MFCC_A, False) if rng.random() < 0.4 else (_lib.MFCC_B, bool(rng.random() < 0.3))
            dev = torch.device("cuda", 0)
            ctx.use_torch_stream()
            span = (n_each - 1) * step + 1024
            raws = [torch.from_numpy(np.clip(rng.normal(0, 10.0 ** rng.uniform(0.5, 4.3), span + 2), -32768, 32767).astype(np.int16)).to(dev) for _ in range(nb)]
            odd = int(rng.integers(0, nb))
            audios = [r[1:1 + span] if (b == odd and rng.random() < 0.5) else r[:span] for b, r in enumerate(raws)]
            outs = [torch.zeros((n_each, 13), dtype=torch.float32, device=dev) for _ in range(nb)]
            feats = [torch.zeros((n_each, 13), dtype=torch.int8, device=dev) for _ in range(nb)]
            if kind == "list":
                ctx.mfcc_batches_t(audios, n_each, step, variant, 13, outs=outs, feats=feats, use_log=use_log)
            else:
                v = variant | (_lib.MFCC_USE_LOG if use_log else 0)
                calls = [ctx.mfcc_queue_call(b & 1, audios[b], n_each, step, v, 13, out=outs[b], feat=feats[b]) for b in range(nb)]
                ctx.queues_fork()
                for c in calls:
                    c()
                ctx.queues_join()
            torch.cuda.synchronize()
            for b in range(nb):
                ro = torch.empty((n_each, 13), dtype=torch.float32, device=dev); rf = torch.empty((n_each, 13), dtype=torch.int8, device=dev)
                ctx.mfcc_t(audios[b], n_each, step, variant, 13, out=ro, feat=rf, use_log=use_log)
                torch.cuda.synchronize()
                assert torch.equal(ro, outs[b]) and torch.equal(rf, feats[b]), (kind, nb, n_each, step, b)
            continue
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
