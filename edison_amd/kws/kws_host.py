"""Host keyword-spotting flow on the MI355X -- the counterpart of the *host* leg of the reference's
``audio/edison/kws/kws_on_mcu.py`` (``fileInference`` :273-308, ``frameInference`` :310-401) and of
``audio/edison/train/kws_nnom.py testfile`` (:335-361), with the board leg (UART) replaced by the GPU:

    wav -> int16 -> pad/cut to 32000 samples -> MFCC variant B -> first 13 coefficients
        -> clip(x*1, -128, 127).round() -> int8 [31][13] -> int8 CNN -> softmax int8[10] -> argmax -> keyword

Everything after the pad/cut runs in ONE C-ABI call (``edison_kws_batch``): the features never visit the host
unless asked for.
"""
import sys

import numpy as np

from .. import config as cfg
from ..context import KEYWORDS, default_context


def read_wav(path):
    """16 kHz wav -> int16 samples; float wavs are scaled like kws_on_mcu.py:328-329."""
    import scipy.io.wavfile as wavfile
    in_fs, data = wavfile.read(path)
    if in_fs != cfg.fs:
        raise ValueError("Sample rate of file %d doesn't match %d" % (in_fs, cfg.fs))
    if data.ndim > 1:
        data = data[:, 0]
    if data.dtype == np.float32 or data.dtype == np.float64:
        data = ((2 ** 15 - 1) * data).astype('int16')
    return np.asarray(data, dtype=np.int16)


def pad_or_cut(data, n=cfg.nSamples, mode="zero"):
    """kws_on_mcu.py:287-290 (zero pad, ``fileInference``) / :331-334 (edge pad, ``frameInference``)."""
    if data.shape[0] < n:
        if mode == "edge":
            return np.pad(data, (0, n - data.shape[0]), mode='edge')
        return np.pad(data, (0, n - data.shape[0]))
    return data[:n]


def infer_utterances(audio, ctx=None):
    """audio: int16 [n_utt, >=31744] (or 1-D single utterance). Returns the dict of ``Context.kws``."""
    ctx = ctx or default_context()
    a = np.atleast_2d(np.asarray(audio, dtype=np.int16))
    return ctx.kws(np.ascontiguousarray(a), n_utt=a.shape[0], utt_stride=a.shape[1])


def report(res, i=0, out=sys.stdout):
    """Prediction line in the spirit of kws_on_mcu.report (:148-157): int8 softmax / 127 and the class."""
    np.set_printoptions(precision=3, suppress=True)
    probs = res["softmax"][i].astype(np.float32) / 127.0
    k = int(res["argmax"][i])
    print('gpu prediction:', probs, KEYWORDS[k], file=out)
    print('dense logits  :', res["logits"][i], file=out)
    return KEYWORDS[k]


def file_inference(path, pad_mode="zero", ctx=None, verbose=True):
    data = pad_or_cut(read_wav(path), mode=pad_mode)
    res = infer_utterances(data, ctx)
    res["keyword"] = KEYWORDS[int(res["argmax"][0])]
    if verbose:
        print('net input (int8, 31x13):')
        print(res["feat"].reshape(cfg.n_frames, cfg.num_mfcc))
        report(res)
    return res


def main(argv):
    """``kws mcu <mode> [file]`` of the reference's CLI (main.py:146-165); modes that only make sense with the
    STM32 board attached (single/mic/hil) report that the board transport is out of scope."""
    if len(argv) < 2:
        print('usage: kws mcu <file|fileinf|frame> <wav>')
        return 1
    mode = argv[1]
    if mode in ("file", "fileinf", "frame", "host"):
        if len(argv) < 3:
            print('need a wav file')
            return 1
        file_inference(argv[2], pad_mode="edge" if mode == "frame" else "zero")
        return 0
    print('mode %r needs the STM32 board (UART host interface), which this port replaces by the GPU' % mode)
    return 1
