#!/usr/bin/env python3
"""Golden vectors for OTHER geometries than audio/config.py's, from the REFERENCE ITSELF (build container only).

mfcc_geom_golden.npz: the reference's `mfcc` (variant A, mfcc_utils.py:134), `mfcc_mcu` (variant B, :255, with and without use_log) and
`batch_mfcc` (:75) imported from /root/reference/audio/edison/mfcc/mfcc_utils.py and run with frame lengths / hops / numbers of mel
bins / filterbank edges / matrix scales that no caller in the reference uses -- powers of two and not (numpy.fft.fft takes any length)
-- on the reference's own wav (audio/data/edison_16k_16b.wav) and seeded noise. Pins the generality path (edison_mfcc_generic).
Inputs and expected outputs only. Re-run:  python3 tests/golden/gen_fixtures_geom.py
"""
import os
import sys

import numpy as np
import scipy.io.wavfile as wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_AUDIO = "/root/reference/audio"
sys.path.insert(0, REF_AUDIO)
import edison.mfcc.mfcc_utils as mfu         # noqa: E402  (reference implementation)

# (name, frame_len, frame_step, mel_nbins, lower Hz, upper Hz, mel_mtx_scale)
CONFIGS = [("g512", 512, 256, 20, 125.0, 3800.0, 128), ("g400", 400, 160, 40, 20.0, 4000.0, 64), ("g2048", 2048, 1024, 64, 80.0, 7600.0, 128),
           ("g1024m26", 1024, 1024, 26, 20.0, 4000.0, 128), ("g1000", 1000, 500, 32, 80.0, 7600.0, 128), ("g33", 33, 7, 5, 300.0, 6000.0, 16)]


def stack(o, key):
    return np.array([np.asarray(f[key]) for f in o])


def main():
    fs, edison = wavfile.read(os.path.join(REF_AUDIO, "data/edison_16k_16b.wav"))
    assert fs == 16000 and edison.dtype == np.int16
    rng = np.random.default_rng(77)
    noise = np.clip(rng.normal(0, 3000, 6000), -32768, 32767).astype(np.int16)
    out = dict(in_edison=edison.astype(np.int16), in_noise=noise,
               configs=np.array([[c[1], c[2], c[3], c[4], c[5], c[6]] for c in CONFIGS], dtype=np.float64), names=np.array([c[0] for c in CONFIGS]))
    for name, N, step, nm, lo, hi, scale in CONFIGS:
        for sname, x in (("edison", out["in_edison"]), ("noise", noise)):
            a = mfu.mfcc(x, fs, len(x), N, step, 0, N, nm, lo, hi)
            b = mfu.mfcc_mcu(x, fs, len(x), N, step, 0, N, nm, lo, hi, scale)
            bl = mfu.mfcc_mcu(x, fs, len(x), N, step, 0, N, nm, lo, hi, scale, True)
            k = "%s_%s_" % (name, sname)
            out[k + "A_mfcc"] = stack(a, "mfcc")
            out[k + "B_mfcc"] = stack(b, "mfcc")
            out[k + "Blog_mfcc"] = stack(bl, "mfcc")
            # the stages of two frames
            for f in (0, len(a) - 1):
                out[k + "A_fft_%d" % f] = np.asarray(a[f]["fft"])
                out[k + "A_spec_%d" % f] = np.asarray(a[f]["spectrogram"])
                out[k + "A_mel_%d" % f] = np.asarray(a[f]["mel_spectrogram"])
                out[k + "A_logmel_%d" % f] = np.asarray(a[f]["log_mel_spectrogram"])
                out[k + "B_fft_%d" % f] = np.asarray(b[f]["fft"])
                out[k + "B_spec_%d" % f] = np.asarray(b[f]["spectrogram"])
                out[k + "B_mel_%d" % f] = np.asarray(b[f]["mel_spectrogram"])
        rows = np.stack([noise[:3000], noise[3000:6000]])
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            out[name + "_batch"] = mfu.batch_mfcc(rows, fs, 3000, N, step, 0, N, nm, lo, hi)
    path = os.path.join(HERE, "mfcc_geom_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
