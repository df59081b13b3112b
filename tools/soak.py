#!/usr/bin/env python3
"""Soak run on the GPU box: minutes of create / load / specialise / run / stream / destroy cycles with every result checked
(oracle on samples, determinism on the large batches), device-memory and host-RSS watched for leaks.
usage (box): tools/soak.py [seconds=300]     prints a progress line every ~20 s and a summary"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from edison_amd import _lib, nnom_import  # noqa: E402
from edison_amd.context import Context  # noqa: E402
from edison_amd.stream import Stream  # noqa: E402
from oracle import net_ref, oracle  # noqa: E402

NAMES = ["same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"]


def blob_of(name):
    with open(os.path.join(ROOT, "tests", "golden", "alt_models", name + ".h")) as f:
        shape, layers = nnom_import.parse_weights_h(f.read())
    return nnom_import.build_blob(shape, layers)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    os.environ["EDISON_JIT_CACHE"] = "/tmp/edison_soak_jit"
    rng = np.random.default_rng(2026)
    oracle.build()
    model = oracle.Model()
    blobs = {n: blob_of(n) for n in NAMES}
    dev = torch.device("cuda", 0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = last = time.time()
    n = dict(contexts=0, loads=0, own_kernels=0, net_inputs=0, stream_pushes=0, kws_utterances=0, mfcc_frames=0, q15_frames=0, calibrations=0, list_launches=0,
             queue_calls=0, generic_frames=0)
    big = Context(0)
    big.use_torch_stream()
    audio = torch.from_numpy(np.clip(rng.normal(0, 3000, 16384 * 31744), -32768, 32767).astype(np.int16)).to(dev)
    ref_am = None
    while time.time() - t0 < seconds:
        # ---- A: a fresh context, a random graph, general kernel then its own, both against the restatement
        name = NAMES[int(rng.integers(0, len(NAMES)))]
        os.environ["EDISON_NET_SPECIALIZE"] = "0"
        c = Context(0, model_path=None)
        c.load_model_bytes(blobs[name])
        info = c.net_info()
        x = rng.integers(-128, 128, (int(rng.integers(1, 700)), info["in_h"] * info["in_w"] * info["in_c"])).astype(np.int8)
        want = net_ref.run(blobs[name], x[:48])
        g = c.net(x)
        assert np.array_equal(g["logits"][:48], want["logits"]) and np.array_equal(g["argmax"][:48], want["argmax"]), ("general", name)
        c.net_specialize()
        o = c.net(x)
        assert np.array_equal(o["logits"], g["logits"]) and np.array_equal(o["argmax"], g["argmax"]), ("own", name)
        c.close()
        n["contexts"] += 1; n["loads"] += 1; n["own_kernels"] += 1; n["net_inputs"] += 2 * x.shape[0]
        # ---- B: a stream of random shape on the shipped model, host and device pushes, against the batch path + oracle CNN
        hop, chunk = 2 * int(rng.integers(64, 513)), int(rng.choice([1, 2, 5, 16, 64]))
        n_push = int(rng.integers(3, 20))
        a = np.clip(rng.normal(0, 2500, n_push * chunk * hop), -32768, 32767).astype(np.int16)
        st = Stream(big, hop=hop, chunk_frames=chunk)
        soft = []
        a_dev = torch.from_numpy(a).to(dev)
        for i in range(n_push):
            sl = slice(i * chunk * hop, (i + 1) * chunk * hop)
            if rng.integers(0, 2):
                so = torch.zeros((chunk, 10), dtype=torch.int8, device=dev)
                st.push_t(a_dev[sl], softmax=so)
                torch.cuda.synchronize()
                soft.append(so.cpu().numpy())
            else:
                soft.append(st.push(a[sl])["softmax"])
        st.close()
        full = np.concatenate([np.zeros(1024 - hop, np.int16), a])
        _, feat = big.mfcc(full, n_frames=n_push * chunk, frame_step=hop, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
        padded = np.concatenate([np.zeros((30, 13), np.int8), feat])
        win = np.stack([padded[i:i + 31].reshape(-1) for i in range(n_push * chunk)])
        assert np.array_equal(np.concatenate(soft), oracle.cnn(model, win, n_threads=4)["softmax"]), ("stream", hop, chunk)
        n["stream_pushes"] += n_push
        # ---- C: a large KWS batch, twice the same answer; float and Q15 MFCC launches
        nu = 16384
        feat_t = torch.zeros((nu, 403), dtype=torch.int8, device=dev)
        lo = torch.zeros((nu, 10), dtype=torch.int8, device=dev)
        am = torch.zeros((nu,), dtype=torch.int32, device=dev)
        big.kws_t(audio, nu, 31744, feat=feat_t, logits=lo, argmax=am, q15=bool(n["loads"] & 1))
        torch.cuda.synchronize()
        key = (bool(n["loads"] & 1), am.cpu().numpy().copy())
        if ref_am is None:
            ref_am = {}
        if key[0] in ref_am:
            assert np.array_equal(ref_am[key[0]], key[1]), "a KWS batch changed its answer"
        ref_am[key[0]] = key[1]
        n["kws_utterances"] += nu
        n["q15_frames" if key[0] else "mfcc_frames"] += nu * 31
        # ---- D (round 5): a fresh context calibrates its two queues (five streams, events: created and destroyed with the context), runs a list of
        # batches in one launch and the same batches over its queues -- both bit-identical to one call per batch; a random other geometry
        # through the generality kernel against numpy
        c = Context(0, model_path=None)
        c.use_torch_stream()
        nb, n_each = int(rng.integers(2, 20)), int(rng.integers(1, 3000))
        xs = [audio[b * n_each * 1024:(b + 1) * n_each * 1024] for b in range(nb)]
        cal = c.queues_calibrate(audio[:8192 * 1024], 8192)
        refs = []
        for b in range(nb):
            o = torch.empty((n_each, 13), dtype=torch.float32, device=dev)
            c.mfcc_t(xs[b], n_each, 1024, _lib.MFCC_B, 13, out=o)
            refs.append(o)
        outs = [torch.zeros((n_each, 13), dtype=torch.float32, device=dev) for _ in range(nb)]
        c.mfcc_batches_t(xs, n_each, 1024, _lib.MFCC_B, 13, outs=outs)
        torch.cuda.synchronize()
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs, refs)), ("list", nb, n_each)
        outs = [torch.zeros((n_each, 13), dtype=torch.float32, device=dev) for _ in range(nb)]
        calls = [c.mfcc_queue_call(b & 1, xs[b], n_each, 1024, _lib.MFCC_B, 13, out=outs[b]) for b in range(nb)]
        c.queues_fork()
        for cl in calls:
            cl()
        c.queues_join()
        torch.cuda.synchronize()
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs, refs)), ("queues", nb, n_each, cal)
        N, nm = int(rng.integers(4, 1200)), int(rng.integers(1, 60))
        gx = np.clip(rng.normal(0, 3000, 3 * N), -32768, 32767).astype(np.int16)
        got = np.zeros((3, nm))
        assert _lib.lib().edison_mfcc_generic(c._h, gx.ctypes.data, 3, N, N, _lib.MFCC_B, nm, 16000.0, 80.0, 7600.0, 128.0, None, None, None, None, got.ctypes.data, 0, None, 1.0) == 0
        ref = oracle.mfcc_numpy(gx, oracle.VARIANT_B, N, N, n_frames=3, num_mel_bins=nm)
        assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), ("generic", N, nm)
        c.close()
        del refs, outs, calls, xs, cl, o
        n["contexts"] += 1; n["calibrations"] += 1; n["list_launches"] += 1; n["queue_calls"] += nb; n["generic_frames"] += 3
        if time.time() - last > 20:
            last = time.time()
            torch.cuda.synchronize()
            print("t=%4.0f s  %s  device memory in use +%.1f MB  host RSS +%.1f MB" % (
                last - t0, n, (free0 - torch.cuda.mem_get_info()[0]) / 1e6, (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0) / 1e3), flush=True)
    big.close()
    del audio, feat_t, lo, am, a_dev
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    grown = (free0 - torch.cuda.mem_get_info()[0]) / 1e6
    print("soak: %.0f s, %s; every result checked; device memory after close: %+.1f MB against the start, host RSS peak +%.1f MB" % (
        time.time() - t0, n, grown, (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0) / 1e3))
    # the ROCm runtime keeps pools it grew once (graph memory, kernel-argument and scratch pools, torch's context): what counts is
    # that the figure printed every 20 s stops moving, which the progress lines show; a leak per cycle would be GBs by now
    assert grown < 512, "device memory grew over the run"


if __name__ == "__main__":
    main()
