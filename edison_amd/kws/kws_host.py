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


def report(res, i=0, out=None):
    """Prediction line in the spirit of kws_on_mcu.report (:148-157): int8 softmax / 127 and the class."""
    out = out or sys.stdout  # looked up per call: a default bound at import time may be a stream that is closed by now
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


def rmse(a, b):
    return np.sqrt(np.mean((a - b) ** 2))


def compare(data_a, data_b, name, out=None):
    """The comparison block of kws_on_mcu.compare (:159-168), same wording and number formats."""
    dev = 100.0 * (1.0 - (data_b.ravel() + 1e-9) / (data_a.ravel() + 1e-9))
    out = out or sys.stdout
    print('_________________________________________________________________', file=out)
    print('Comparing: %s' % (name), file=out)
    print("Deviation: max %.3f%% min %.3f%% avg %.3f%% \nrmse %.3f" % (
        dev.max(), dev.min(), np.mean(dev), rmse(data_b.ravel(), data_a.ravel())), file=out)
    print('scale %.3f=1/%.3f' % (data_b.max() / data_a.max(), data_a.max() / data_b.max()), file=out)
    print('correlation coeff %.3f' % (np.corrcoef(data_a.ravel(), data_b.ravel())[0, 1]), file=out)
    print('_________________________________________________________________', file=out)


def frame_inference(path, ctx=None, out=None):
    """`kws mcu file <wav>` = kws_on_mcu.frameInference (:310-401): the wav (edge-padded to 2 s) through the host
    leg (MFCC variant B) and through the board's leg -- here the GPU's variant C, i.e. the firmware's own Q15
    arithmetic -- each followed by the int8 network, then the two comparison blocks the reference prints
    (README.md:121-139 shows them for data/edison_16k_16b.wav)."""
    out = out or sys.stdout
    from .. import _lib
    ctx = ctx or default_context()
    data = pad_or_cut(read_wav(path), mode="edge")
    np.set_printoptions(precision=3, suppress=True)
    host = ctx.kws(data, n_utt=1, utt_stride=data.shape[0])
    mcu = ctx.kws(data, n_utt=1, utt_stride=data.shape[0], q15=True)
    host_pred = host["softmax"][0].astype(np.float32) / 127.0
    mcu_pred = mcu["softmax"][0].astype(np.float32) / 127.0
    print('keywords:', list(KEYWORDS), file=out)
    print('host prediction:', host_pred, KEYWORDS[int(host["argmax"][0])], file=out)
    print('mcu prediction: ', mcu_pred, KEYWORDS[int(mcu["argmax"][0])], file=out)
    print('rmse:', rmse(host_pred, mcu_pred), file=out)
    compare(host_pred, mcu_pred, 'predictions', out)
    # the MFCC block compares the unclipped coefficients, float32 like the reference's arrays (:346, app.c:212)
    host_mfcc = ctx.mfcc(data, variant=_lib.MFCC_B, n_coef=cfg.num_mfcc).astype(np.float32)
    mcu_mfcc = ctx.mfcc_q15(data, n_coef=cfg.num_mfcc).astype(np.float32)
    compare(host_mfcc, mcu_mfcc, 'MFCC=net input', out)
    return dict(host=host, mcu=mcu, host_mfcc=host_mfcc, mcu_mfcc=mcu_mfcc)


def single_inference(repeat=1, ctx=None, out=None):
    """`kws mcu single [n]` = kws_on_mcu.singleInference (:236-270), nnom branch: the all-zero int8 net input."""
    out = out or sys.stdout
    ctx = ctx or default_context()
    res = None
    for _ in range(max(1, int(repeat))):
        res = ctx.cnn(np.zeros((1, cfg.n_frames * cfg.num_mfcc), np.int8))
        report(res, 0, out)
    return res


# the reference's own function names and call conventions (kws_on_mcu.py:243,273,310: `args` = the CLI's remaining
# arguments, args[0] = the wav file)
def singleInference(repeat=1):
    return single_inference(repeat)


def fileInference(args):
    return file_inference(args[0])


def frameInference(args):
    return frame_inference(args[0])


def main(argv):
    """``kws mcu <mode> [file]`` of the reference's CLI (main.py:146-165, kws_on_mcu.py:650-690). The modes that record
    from a microphone (mic, host, hostcont, hostsingle, miccont) are not part of this port."""
    if len(argv) < 2:
        print('usage: kws mcu <single [n] | fileinf <wav> | file <wav> | frame <wav>>')
        return 1
    mode = argv[1]
    print('Running mode', mode, 'with args', argv[2:])
    if mode == "single":
        single_inference(int(argv[2]) if len(argv) > 2 else 1)
        return 0
    if mode in ("file", "fileinf", "frame", "host"):
        if len(argv) < 3:
            print('need a wav file')
            return 1
        if mode == "file":
            frame_inference(argv[2])
        else:
            file_inference(argv[2], pad_mode="edge" if mode == "frame" else "zero")
        return 0
    print('mode %r records from a microphone, which this port does not drive' % mode)
    return 1
