#!/usr/bin/env python3
"""Random (hop, chunk, variant) streams against the batch path and the oracle on the GPU box: every output of a push
must equal the batch result on the corresponding 31-frame window (tests/test_gpu_stream.py holds four fixed
shapes).   usage (box): tools/fuzz_stream.py [n [seed]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from edison_amd import _lib  # noqa: E402
from edison_amd.context import Context  # noqa: E402
from edison_amd.stream import Stream  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
    oracle.build()
    ctx = Context(0)
    model = oracle.Model()
    for case in range(n_cases):
        hop = 2 * int(rng.integers(1, 513))
        chunk = int(rng.choice([1, 2, 3, 5, 8, 13, 31, 40, 100]))
        q15 = bool(rng.integers(0, 2))
        n_push = int(rng.integers(1, 6)) if chunk > 20 else int(rng.integers(2, 40))
        n_frames = n_push * chunk
        audio = np.clip(rng.normal(0, 10.0 ** rng.uniform(1.5, 4.0), n_frames * hop), -32768, 32767).astype(np.int16)
        st = Stream(ctx, hop=hop, chunk_frames=chunk, q15=q15)
        outs = [st.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
        st.close()
        soft = np.concatenate([o["softmax"] for o in outs])
        am = np.concatenate([o["argmax"] for o in outs])
        full = np.concatenate([np.zeros(1024 - hop, np.int16), audio])
        if q15:
            _, feat = ctx.mfcc_q15(full, n_frames=n_frames, frame_step=hop, n_coef=13, want_feat=True)
            assert np.array_equal(feat, oracle.net_input_q15(oracle.mfcc_q15(full, n_frames=n_frames, frame_step=hop)[:, :13], n_coef=13)), ("q15 feat", hop, chunk)
        else:
            _, feat = ctx.mfcc(full, n_frames=n_frames, frame_step=hop, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
        padded = np.concatenate([np.zeros((30, 13), np.int8), feat])
        win = np.stack([padded[i:i + 31].reshape(-1) for i in range(n_frames)])
        o = oracle.cnn(model, win, n_threads=4)
        assert np.array_equal(soft, o["softmax"]) and np.array_equal(am, o["argmax"]), (hop, chunk, q15, n_push)
    print("%d random streams equal the batch path window by window and the oracle's CNN bit for bit" % n_cases)


if __name__ == "__main__":
    main()
