"""Host-side mirror of the table generator in the reference's ``audio/edison/mfcc/mfcc_on_mcu.py``.

The firmware's MFCC (variant C) reads its constants from ``firmware/src/audio/mel_constants.h``, which the
reference generates with ``calcCConstants()`` (mfcc_on_mcu.py:68-145): the dense int16 mel matrix, a log LUT, the
DCT twiddle factors of Makhoul's method, and the compact (non-sparse) mel matrix with per-band start / count
(``melMtxToUnspares``, :26-66). This module produces the same header text from the same parameters, so that a
firmware build and the GPU's variant C (``csrc/tables_q15.c``, which derives its compact tables the same way) can
be regenerated together for another filterbank. The mel matrix comes from the library's host-side
``edison_gen_mel_weight_matrix``; everything else is numpy on a few thousand numbers.

  melMtxToUnspares   mfcc_on_mcu.py:26-66
  calcCConstants     mfcc_on_mcu.py:68-145   (returns the text; writes it when a file name is given)

The C-array formatting follows ``mcu_util.vecToC`` / ``mtxToC`` (mcu_util.py:548-577): right-aligned ``%{prepad}d``
fields, a line break once a line has reached 80 characters, matrices one row per line.
"""
import numpy as np

from .. import config as _cfg
from . import mfcc_utils as mfu


def _c_vector(values, prepad=3, maxwidth=80):
    """``{  1,  2,  3}`` with the reference's wrapping rule: break before an element once the current line is
    already ``maxwidth`` characters long (0 = never)."""
    lines, cur = [], "{"
    for v in values:
        if maxwidth and len(cur) >= maxwidth:
            lines.append(cur)
            cur = ""
        cur += "%*d," % (prepad, int(v))
    lines.append(cur)
    return "\n".join(lines)[:-1] + "}"


def _c_matrix(matrix, prepad=3):
    return "{ \n" + ",\n".join("  " + _c_vector(row, prepad, 0) for row in matrix) + "\n}"


def melMtxToUnspares(mel_mtx):
    """Dense [bins, bands] integer mel matrix -> (melMtxCompact, melCompFStarts, melCompFCount): per band, `count`
    (= number of non-zero entries) consecutive coefficients from the first non-zero bin on."""
    compact, starts, counts = [], [], []
    for band in np.asarray(mel_mtx).T:
        nz = np.flatnonzero(band)
        first, count = int(nz[0]), int(nz.size)
        compact.extend(band[first:first + count].tolist())
        starts.append(first)
        counts.append(count)
    compact = np.array(compact, dtype="int16")
    assert int(np.sum(counts)) == compact.size
    return compact, np.array(starts, dtype="int16"), np.array(counts, dtype="int16")


def calcCConstants(fname=None, sample_size=_cfg.frame_length, num_mel_bins=_cfg.num_mel_bins,
                   num_spectrogram_bins=_cfg.num_spectrogram_bins, sample_rate=_cfg.fs,
                   lower_edge_hertz=_cfg.lower_edge_hertz, upper_edge_hertz=_cfg.upper_edge_hertz,
                   mel_mtx_scale=_cfg.mel_mtx_scale, mel_twiddle_scale=_cfg.mel_twiddle_scale):
    """The text of mel_constants.h for these parameters (and the file, when `fname` is given)."""
    mel_mtx = mfu.gen_mel_weight_matrix(num_mel_bins=num_mel_bins, num_spectrogram_bins=num_spectrogram_bins,
                                        sample_rate=sample_rate, lower_edge_hertz=lower_edge_hertz,
                                        upper_edge_hertz=upper_edge_hertz)
    mel_mtx_s16 = np.array(mel_mtx_scale * mel_mtx, dtype="int16")          # truncation, like the reference
    # ln(x) for x in [0, 32766] as int16 (the firmware keeps the table although step [4.] is commented out)
    log_lut = np.array(np.log(np.linspace(1e-6, 32766, 32767)), dtype="int16")
    # scale vector of the fast DCT (Makhoul 1980): 2 exp(-j pi k / 2N), interleaved re / im
    k = np.arange(num_mel_bins)
    factors = 2 * np.exp(-1j * np.pi * k / (2 * num_mel_bins))
    tw = np.empty(2 * num_mel_bins, dtype="int16")
    tw[0::2] = np.array(mel_twiddle_scale * factors.real, dtype="int16")
    tw[1::2] = np.array(mel_twiddle_scale * factors.imag, dtype="int16")
    compact, starts, counts = melMtxToUnspares(mel_mtx_s16)

    def define(name, value, width=32):
        return "#define %s%s\n" % (name.ljust(width - 8), value)

    text = "".join([
        define("MEL_SAMPLE_SIZE", "%5d" % sample_size),
        define("MEL_N_MEL_BINS", "%5d" % num_mel_bins),
        define("MEL_N_SPECTROGRAM_BINS", "%5d" % num_spectrogram_bins),
        define("MEL_SAMPLE_RATE", "%5d" % sample_rate),
        define("MEL_LOWER_EDGE_HZ", "%05.3f" % lower_edge_hertz),
        define("MEL_UPPER_EDGE_HZ", "%05.3f" % upper_edge_hertz),
        define("MEL_MTX_SCALE", "%5d" % mel_mtx_scale) + "\n",
        define("MEL_MTX_ROWS", "%5d" % mel_mtx_s16.shape[0]),
        define("MEL_MTX_COLS", "%5d" % mel_mtx_s16.shape[1]),
        "const int16_t melMtx[%d][%d] = \n%s;\n" % (mel_mtx_s16.shape[0], mel_mtx_s16.shape[1], _c_matrix(mel_mtx_s16, 4)),
        define("MEL_LOG_LUT_SIZE", "%5d" % log_lut.shape[0]),
        "const q15_t logLutq15[%d] = \n%s;\n" % (log_lut.shape[0], _c_vector(log_lut, 4)),
        define("MEL_DCT_TWIDDLE_SIZE", "%5d" % tw.shape[0], 36),
        "const q15_t dctTwiddleFactorsq15[%d] = \n%s;\n" % (tw.shape[0], _c_vector(tw, 4)),
        "\n\n// Compact mel matrix\n",
        "const q15_t melMtxCompact[%d] = \n%s;\n" % (compact.size, _c_vector(compact, 4)),
        "const q15_t melCompFStarts[%d] = \n%s;\n" % (starts.size, _c_vector(starts, 4)),
        "const q15_t melCompFCount[%d] = \n%s;\n" % (counts.size, _c_vector(counts, 4)),
    ])
    if fname:
        with open(fname, "w") as f:
            f.write(text)
    return text


# ---------------------------------------------------------------------------------------------------------------
# The hardware-in-the-loop comparisons of the reference script, with the board's leg (audioCalcMFCCs over the UART,
# audioDumpToHost: audioprocessing.c:221-231) computed by the GPU's variant C -- the same integers, no board.

def _two_tone():
    """The synthetic frame of mfcc_on_mcu.py:312-315."""
    t = np.linspace(0, _cfg.frame_length / _cfg.fs, _cfg.frame_length)
    return np.array(1000 * np.cos(2 * np.pi * (_cfg.fs / 16) * t) + 500 * np.cos(2 * np.pi * (_cfg.fs / 128) * t), dtype="int16")


def _read_int16(path):
    import scipy.io.wavfile as wavfile
    fs, data = wavfile.read(path)
    data = np.asarray(data)
    if data.dtype == np.float32:
        data = np.array((2 ** 15 - 1) * data, dtype="int16")   # mfcc_on_mcu.py:424-425
    return fs, np.asarray(data, dtype=np.int16)


def compare_stages(y, out=None):
    """Variant B on the host model vs variant C ("mcu") for the frames of y; prints the four scale lines of
    mfcc_on_mcu.py:382-392 and returns both sets of stage arrays."""
    import sys
    from ..context import default_context
    from .. import _lib
    out = out or sys.stdout
    ctx = default_context()
    host = ctx.mfcc_stages(y, variant=_lib.MFCC_B)
    mcu = ctx.mfcc_q15_stages(y)
    host_fft = host["fft"] / float(_cfg.frame_length)                       # mfcc_mcu keeps fft / 1024 (mfcc_utils.py:297)
    print("host/mcu fft scale %f" % (np.real(host_fft).max() / mcu["fft"][..., 0].max()), file=out)
    print("host/mcu spectrum scale %f" % (host["spectrogram"].max() / mcu["spectrogram"].max()), file=out)
    print("host/mcu mel spectrum scale %f" % (host["mel_spectrogram"].max() / mcu["mel_spectrogram"].max()), file=out)
    print("host/mcu dct scale %f" % (host["mfcc"].max() / mcu["mfcc"].max()), file=out)
    return host, mcu


def modeSingle(wav=None):
    """`mfcc mcu single`: one 1024-sample frame. With a wav: its first samples padded with 6 like the reference does
    with data/hey_short_16k.wav (mfcc_on_mcu.py:320-323); without: the synthetic two-tone."""
    if wav:
        _, d = _read_int16(wav)
        d = d[:_cfg.frame_length]
        y = np.pad(d, (0, _cfg.frame_length - d.shape[0]), "constant", constant_values=(4, 6)).astype(np.int16)
    else:
        y = _two_tone()
    return compare_stages(y)


def modeCalc(fname="mel_constants.h"):
    """`mfcc mcu calc`: write the firmware's constant header."""
    calcCConstants(fname)
    print("wrote %s" % fname)


def modeFile(fname):
    """`mfcc mcu file <wav>`: every 1024-sample frame of the file through both variants (mfcc_on_mcu.py:407-470)."""
    fs, y = _read_int16(fname)
    print("Working with %s" % fname)
    print("Frame length in seconds = %.3fs" % (_cfg.frame_length / fs))
    print("Number of input samples = %d" % len(y))
    host, mcu = compare_stages(y)
    # the coefficients the network sees; beyond index 16 the firmware's RFFT-based "DCT" mirrors its own output
    # (out[32-k] = out[k], audioprocessing.c:330-436) and is not comparable with a DCT-II any more
    nc = _cfg.num_mfcc
    a, b = host["mfcc"][:, :nc].ravel(), mcu["mfcc"][:, :nc].astype(np.float64).ravel()
    print("MFCC[:%d] host vs mcu over %d frames: rmse %.3f, correlation coeff %.3f" % (
        nc, host["mfcc"].shape[0], np.sqrt(np.mean((a - b) ** 2)), np.corrcoef(a, b)[0, 1]))
    return host, mcu


def main(argv):
    """``mfcc mcu <mode>`` of the reference's CLI (mfcc_on_mcu.py:526-548)."""
    if len(argv) < 2:
        print("Usage:\n  mfcc mcu <mode>\n    calc [file]   Calculate C constants header file\n"
              "    single [wav]  Run MFCC on single frame\n    file <wav>    Run MFCC on wav file of any length")
        return 1
    mode = argv[1]
    if mode == "single":
        modeSingle(argv[2] if len(argv) > 2 else None)
    elif mode == "calc":
        modeCalc(argv[2] if len(argv) > 2 else "mel_constants.h")
    elif mode == "file":
        if len(argv) < 3:
            print("Specify input file")
            return 1
        modeFile(argv[2])
    else:
        print("Unrecognized mode")
        return 1
    return 0
