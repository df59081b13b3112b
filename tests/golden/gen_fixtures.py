#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE ITSELF (run in the build container only).

  mfcc_golden.npz : inputs + outputs of the reference's Python host MFCC, obtained by importing
                    /root/reference/audio/edison/mfcc/mfcc_utils.py (variants A `mfcc` :134 and
                    B `mfcc_mcu` :255, `gen_mel_weight_matrix` :36) on
                      - the two wav files the reference's own scripts use (audio/data/edison_16k_16b.wav,
                        mfcc.py:175; audio/data/hey_short_16k.wav padded to 1024 with constant 6,
                        mfcc_on_mcu.py:321-323),
                      - the synthetic two-tone of mfcc_on_mcu.py:314-315,
                      - seeded Gaussian noise, silence, full-scale extremes.
  cnn_golden.npz  : int8 inputs + every layer's activations from the reference int8 CNN
                    (NNoM 0.3.0 + CMSIS-NN + weights.h, compiled by oracle/Makefile into
                    oracle/_ref/libnnom_ref.so), incl. the known-answer inputs (zeros, +127, -128, LCG).
  kws_golden.npz  : the host KWS flow of kws_on_mcu.py:273-401 / kws_nnom.py:335-361 on edison_16k_16b.wav:
                    pad (zero / edge) -> variant B -> [:13] -> clip/round -> int8 -> reference CNN.

The reference cannot travel to the GPU box; these small data files (inputs and expected outputs only)
can. Re-run:  python3 tests/golden/gen_fixtures.py
"""
import os
import sys

import numpy as np
import scipy.io.wavfile as wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_AUDIO = "/root/reference/audio"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF_AUDIO)

import config as refcfg                      # noqa: E402  (reference audio/config.py)
import edison.mfcc.mfcc_utils as mfu         # noqa: E402  (reference implementation)
from oracle import oracle                    # noqa: E402


def ref_mfcc_a(x):
    o = mfu.mfcc(x, refcfg.fs, len(x), refcfg.frame_len, refcfg.frame_step, 0, refcfg.fft_len,
                 refcfg.mel_nbins, refcfg.mel_lower_hz, refcfg.mel_upper_hz)
    return o


def ref_mfcc_b(x, use_log=False):
    o = mfu.mfcc_mcu(x, refcfg.fs, len(x), refcfg.frame_len, refcfg.frame_step, 0, refcfg.fft_len,
                     refcfg.mel_nbins, refcfg.mel_lower_hz, refcfg.mel_upper_hz, refcfg.mel_mtx_scale, use_log)
    return o


def stack(o, key):
    return np.array([np.asarray(f[key]) for f in o])


def main():
    oracle.build()
    out = {}

    # ---------------------------------------------------------------- MFCC inputs
    fs, edison = wavfile.read(os.path.join(REF_AUDIO, "data/edison_16k_16b.wav"))
    assert fs == 16000 and edison.dtype == np.int16
    fs2, hey = wavfile.read(os.path.join(REF_AUDIO, "data/hey_short_16k.wav"))
    assert fs2 == 16000
    if hey.dtype != np.int16:
        hey = ((2 ** 15 - 1) * hey).astype("int16")
    hey_frame = np.pad(hey[:1024], (0, max(0, 1024 - len(hey))), "constant", constant_values=6).astype(np.int16)

    t = np.linspace(0, 1024 / 16000.0, num=1024)
    two_tone = np.array(1000 * np.cos(2 * np.pi * 16000 / 16 * t) + 500 * np.cos(2 * np.pi * 16000 / 128 * t),
                        dtype="int16")  # mfcc_on_mcu.py:314-315
    rng = np.random.default_rng(20)
    noise = np.clip(rng.normal(0, 3000, 31 * 1024), -32768, 32767).astype(np.int16)
    quiet = np.clip(rng.normal(0, 0.01 * 32767, 4 * 1024), -32768, 32767).astype(np.int16)  # 1 %-FS noise class
    extremes = np.concatenate([np.zeros(1024, np.int16), np.full(1024, 32767, np.int16),
                               np.full(1024, -32768, np.int16),
                               np.tile(np.array([32767, -32768], np.int16), 512),
                               (rng.integers(-32768, 32768, 1024)).astype(np.int16)])
    streams = dict(edison=edison.astype(np.int16), hey=hey_frame, two_tone=two_tone, noise=noise, quiet=quiet,
                   extremes=extremes)

    mfcc = {}
    for name, x in streams.items():
        oa, ob = ref_mfcc_a(x), ref_mfcc_b(x)
        mfcc["in_" + name] = x
        mfcc["A_mfcc_" + name] = stack(oa, "mfcc")
        mfcc["B_mfcc_" + name] = stack(ob, "mfcc")
        if name in ("edison", "two_tone", "extremes"):
            mfcc["A_spec_" + name] = stack(oa, "spectrogram")
            mfcc["A_mel_" + name] = stack(oa, "mel_spectrogram")
            mfcc["A_logmel_" + name] = stack(oa, "log_mel_spectrogram")
            mfcc["B_spec_" + name] = stack(ob, "spectrogram")
            mfcc["B_mel_" + name] = stack(ob, "mel_spectrogram")
    mfcc["Blog_mfcc_edison"] = stack(ref_mfcc_b(streams["edison"], use_log=True), "mfcc")
    # batch_mfcc (mfcc_utils.py:75-131): [n, samples] -> [n, frames, 32]
    batch_in = noise[:8 * 1024].reshape(4, 2048)
    mfcc["batch_in"] = batch_in
    mfcc["batch_out"] = mfu.batch_mfcc(batch_in, refcfg.fs, 2048, 1024, 1024, 0, 1024, 32, 80.0, 7600.0)
    # overlapping frames (frame_step 512) through variant B
    ov = mfu.mfcc_mcu(noise[:4096], refcfg.fs, 4096, 1024, 512, 0, 1024, 32, 80.0, 7600.0, 128)
    mfcc["B_mfcc_overlap512"] = stack(ov, "mfcc")
    mfcc["mel_W512"] = mfu.gen_mel_weight_matrix(32, 512, 16000, 80.0, 7600.0)
    mfcc["mel_W513"] = mfu.gen_mel_weight_matrix(32, 513, 16000, 80.0, 7600.0)
    mfcc["mel_W129_20"] = mfu.gen_mel_weight_matrix()  # the function's own defaults
    np.savez_compressed(os.path.join(HERE, "mfcc_golden.npz"), **mfcc)

    # ---------------------------------------------------------------- CNN
    feats = [np.zeros(403, np.int8), np.full(403, 127, np.int8), np.full(403, -128, np.int8)]
    s, lcg = 12345, []
    for _ in range(403):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        lcg.append(np.uint8(s >> 24).astype(np.int8))
    feats.append(np.array(lcg, np.int8))
    r2 = np.random.default_rng(21)
    for i in range(40):
        feats.append(r2.integers(-128, 128, 403).astype(np.int8))           # full-range uniform
    for i in range(12):
        feats.append(np.clip(r2.normal(0, 20, 403), -128, 127).round().astype(np.int8))  # small-amplitude
    # wav-derived features (variant B of the reference -> clip/round like kws_nnom.py:359-361)
    for pad_mode in ("zero", "edge"):
        d = edison
        d = np.pad(d, (0, 32000 - len(d))) if pad_mode == "zero" else np.pad(d, (0, 32000 - len(d)), mode="edge")
        ob = ref_mfcc_b(d.astype(np.int16))
        m = np.array([f["mfcc"][:13] for f in ob])
        ni = np.clip(np.array(m, dtype="float32") * 1.0, -128, 127).round().astype(np.int8)
        feats.append(ni.reshape(-1))
        out["kws_%s_audio" % pad_mode] = d.astype(np.int16)
        out["kws_%s_mfcc" % pad_mode] = m
        out["kws_%s_feat" % pad_mode] = ni
    feats = np.stack(feats)
    names = ["input", "conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "dense", "softmax", "output"]
    layers = {n: [] for n in names}
    for f in feats:
        acts = oracle.nnom_ref_layers(f)
        assert len(acts) == len(names)
        for n, a in zip(names, acts):
            layers[n].append(a)
    cnn = {"feats": feats}
    for n in names:
        cnn[n] = np.stack(layers[n])
    cnn["argmax"] = oracle.nnom_ref_batch(feats)["argmax"]
    np.savez_compressed(os.path.join(HERE, "cnn_golden.npz"), **cnn)

    for pad_mode, idx in (("zero", -2), ("edge", -1)):
        out["kws_%s_logits" % pad_mode] = cnn["dense"][idx]
        out["kws_%s_softmax" % pad_mode] = cnn["softmax"][idx]
        out["kws_%s_argmax" % pad_mode] = cnn["argmax"][idx]
    np.savez_compressed(os.path.join(HERE, "kws_golden.npz"), **out)

    print("mfcc_golden.npz:", {k: v.shape for k, v in mfcc.items() if not k.startswith("in_")})
    print("cnn_golden.npz: feats", feats.shape, "known answers:")
    for i, nm in enumerate(["zeros", "+127", "-128", "lcg"]):
        print("  %-5s dense %s softmax %s argmax %d" % (nm, cnn["dense"][i], cnn["softmax"][i], cnn["argmax"][i]))
    print("kws zero-pad logits", out["kws_zero_logits"], "argmax", out["kws_zero_argmax"])
    print("kws edge-pad logits", out["kws_edge_logits"], "argmax", out["kws_edge_argmax"])
    print("variant A frame 3 mfcc[:3]", mfcc["A_mfcc_edison"][3][:3])
    print("variant B frame 3 mfcc[:3]", mfcc["B_mfcc_edison"][3][:3])


if __name__ == "__main__":
    main()
