#!/usr/bin/env python3
"""Quick on-GPU numbers while developing: HIP path vs oracle on the golden inputs (prints error statistics)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from edison_amd.context import Context  # noqa: E402
from edison_amd import _lib            # noqa: E402
from oracle import oracle              # noqa: E402


def main():
    ctx = Context(0)
    print(ctx.device_info())
    g = np.load(os.path.join(ROOT, "tests/golden/mfcc_golden.npz"))
    for name in ["edison", "hey", "two_tone", "noise", "quiet", "extremes"]:
        x = g["in_" + name]
        for v, k in ((_lib.MFCC_A, "A"), (_lib.MFCC_B, "B")):
            out = ctx.mfcc(x, variant=v, n_coef=32)
            ref = g[k + "_mfcc_" + name]
            d = np.abs(out - ref)
            print("%-9s %s n=%3d  max|d|=%.3e  max|d|/(1+|ref|)=%.3e  max|ref|=%.3e" % (
                name, k, len(ref), d.max(), (d / (1 + np.abs(ref))).max(), np.abs(ref).max()))
    st = ctx.mfcc_stages(g["in_edison"], variant=_lib.MFCC_A)
    for key, gk in (("spectrogram", "A_spec_edison"), ("mel_spectrogram", "A_mel_edison"),
                    ("log_mel_spectrogram", "A_logmel_edison")):
        ref = g[gk]
        got = st[key][:, :ref.shape[1]]
        print("stage A %-20s max rel-to-max err %.3e" % (key, np.abs(got - ref).max() / np.abs(ref).max()))
    st = ctx.mfcc_stages(g["in_edison"], variant=_lib.MFCC_B)
    ref = g["B_spec_edison"][:, :513]
    print("stage B spectrogram          max rel-to-max err %.3e" % (np.abs(st["spectrogram"] - ref).max() / np.abs(ref).max()))
    fftref = np.fft.fft(g["in_edison"][:10240].reshape(10, 1024).astype(np.float64))[:, :513]
    print("fft max rel-to-max err %.3e" % (np.abs(st["fft"] - fftref).max() / np.abs(fftref).max()))

    c = np.load(os.path.join(ROOT, "tests/golden/cnn_golden.npz"))
    r = ctx.cnn(c["feats"])
    print("cnn golden: logits", (r["logits"] == c["dense"]).all(), "softmax", (r["softmax"] == c["softmax"]).all(),
          "argmax", (r["argmax"] == c["argmax"]).all())
    lay = ctx.cnn_layers(c["feats"])
    for k in lay:
        print("  layer %-8s %s" % (k, (lay[k] == c[k]).all()))
    rng = np.random.default_rng(7)
    f = rng.integers(-128, 128, (20000, 403)).astype(np.int8)
    M = oracle.Model()
    t = time.time(); ro = oracle.cnn(M, f, n_threads=8); t_cpu = time.time() - t
    t = time.time(); rg = ctx.cnn(f); t_gpu = time.time() - t
    print("cnn random 20000: logits", (rg["logits"] == ro["logits"]).all(), "softmax", (rg["softmax"] == ro["softmax"]).all(),
          "argmax", (rg["argmax"] == ro["argmax"]).all(), "cpu %.2fs gpu(host ptr) %.3fs" % (t_cpu, t_gpu))

    k = np.load(os.path.join(ROOT, "tests/golden/kws_golden.npz"))
    for mode in ("zero", "edge"):
        r = ctx.kws(k["kws_%s_audio" % mode], n_utt=1)
        nflip = int((r["feat"].reshape(31, 13) != k["kws_%s_feat" % mode]).sum())
        print("kws %s: feature flips %d/403, logits %s (ref %s) argmax %d (ref %d)" % (
            mode, nflip, r["logits"][0], k["kws_%s_logits" % mode], r["argmax"][0], k["kws_%s_argmax" % mode]))


if __name__ == "__main__":
    main()
