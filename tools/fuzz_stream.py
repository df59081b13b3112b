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
    dev = torch.device("cuda", 0)
    ctx.use_torch_stream()
    dev_pushes = 0
    for case in range(n_cases):
        hop = 2 * int(rng.integers(1, 513))
        chunk = int(rng.choice([1, 2, 3, 5, 8, 13, 31, 40, 100]))
        q15 = bool(rng.integers(0, 2))
        n_push = int(rng.integers(1, 6)) if chunk > 20 else int(rng.integers(2, 40))
        n_frames = n_push * chunk
        audio = np.clip(rng.normal(0, 10.0 ** rng.uniform(1.5, 4.0), n_frames * hop), -32768, 32767).astype(np.int16)
        st = Stream(ctx, hop=hop, chunk_frames=chunk, q15=q15)
        # how the samples arrive: host pushes, device pushes (they slide through the history buffers and wrap every eighth
        # push), or a random mixture (the history moves between host and device and back to the front of its buffers)
        how = int(rng.integers(0, 3))
        dev_pushes += how > 0
        a_dev = torch.from_numpy(audio).to(dev)
        so = torch.zeros((n_frames, 10), dtype=torch.int8, device=dev)
        amd = torch.zeros((n_frames,), dtype=torch.int32, device=dev)
        outs = []
        for i in range(n_push):
            sl = slice(i * chunk * hop, (i + 1) * chunk * hop)
            if how == 1 or (how == 2 and rng.integers(0, 2)):
                st.push_t(a_dev[sl], softmax=so[i * chunk:(i + 1) * chunk], argmax=amd[i * chunk:(i + 1) * chunk])
                outs.append(None)
            else:
                outs.append(st.push(audio[sl]))
        torch.cuda.synchronize()
        st.close()
        so, amd = so.cpu().numpy(), amd.cpu().numpy()
        soft = np.concatenate([so[i * chunk:(i + 1) * chunk] if o is None else o["softmax"] for i, o in enumerate(outs)])
        am = np.concatenate([amd[i * chunk:(i + 1) * chunk] if o is None else np.asarray(o["argmax"]).reshape(-1) for i, o in enumerate(outs)])
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
    print("%d random streams (%d of them fed wholly or partly through device pushes) equal the batch path window by window and the oracle's CNN bit for bit" % (n_cases, dev_pushes))


if __name__ == "__main__":
    main()
