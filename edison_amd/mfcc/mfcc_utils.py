"""Host-side mirror of the reference's ``audio/edison/mfcc/mfcc_utils.py`` -- same function names, positional
arguments, return structures and quirks, with the per-frame arithmetic done by the HIP kernels behind the
C-ABI (``libedison_hip.so``) instead of numpy/scipy. Callers of the reference (mfcc.py:188-190,
kws_keras.py:450, kws_on_mcu.py:293,343,...) can import this module in its place.

  frames                  mfcc_utils.py:16-27     (pure indexing, stays on the host)
  hertz_to_mel            mfcc_utils.py:30-34
  gen_mel_weight_matrix   mfcc_utils.py:36-73     -> edison_gen_mel_weight_matrix (host C, float64)
  batch_mfcc              mfcc_utils.py:75-131    -> edison_mfcc_batch, variant A
  mfcc                    mfcc_utils.py:134-199   -> edison_mfcc_stages, variant A
  mfcc_mcu                mfcc_utils.py:255-323   -> edison_mfcc_stages, variant B

  dct2Makhoul             mfcc_utils.py:324-343   (host numpy helper of the board tools: DCT-II through one FFT)
  mfcc_tf                 mfcc_utils.py:201-253   -> edison_mfcc_stages, variant TF: tf.signal's pipeline (periodic Hann
                                                  window, rfft, 513-bin mel matrix, ln, dct2*rsqrt(64)) on the GPU.
                                                  TensorFlow is not in this image: PARITY UNPINNED, checked against a
                                                  float64 restatement of tf.signal's published definitions only

The fast GPU kernels are specialised for the reference's shipped geometry: 1024-sample frames and 32 mel bins
(audio/config.py:15,19) -- what every caller in the reference passes; their arrays are float32 promoted to float64 (the
reference computes in float64; DESIGN.md has the measured tolerance). Any OTHER geometry the reference's functions accept
(frame_len 4 .. 4096, power of two or not; mel_nbins 1 .. 256) goes through the generality kernel (`edison_mfcc_generic`:
float64 on the GPU, one workgroup per frame, the reference's values to ~1e-12; round 5; mfcc_tf too, with fft_len == frame_len). Outside those limits:
NotImplementedError, never another path.
"""
import ctypes

import numpy as np

from .. import _lib
from .. import config as _cfg
from ..context import default_context

MEL_HIGH_FREQUENCY_Q = _cfg.MEL_HIGH_FREQUENCY_Q
MEL_BREAK_FREQUENCY_HERTZ = _cfg.MEL_BREAK_FREQUENCY_HERTZ


def frames(data, frame_length=3, frame_step=1):
    """Split a data vector into (possibly overlapping) frames (mfcc_utils.py:16-27)."""
    data = np.asarray(data)
    n_frames = 1 + (data.shape[0] - frame_length) // frame_step
    out = np.zeros((n_frames, frame_length))
    for i in range(n_frames):
        out[i] = data[i * frame_step:i * frame_step + frame_length]
    return out


def hertz_to_mel(frequencies_hertz):
    """Hertz -> mel (mfcc_utils.py:30-34)."""
    return MEL_HIGH_FREQUENCY_Q * np.log(1.0 + (frequencies_hertz / MEL_BREAK_FREQUENCY_HERTZ))


def gen_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                          lower_edge_hertz=125.0, upper_edge_hertz=3800.0):
    """[num_spectrogram_bins, num_mel_bins] float64 triangular mel weights, DC row zero (mfcc_utils.py:36-73)."""
    W = np.zeros((int(num_spectrogram_bins), int(num_mel_bins)), np.float64)
    r = _lib.lib().edison_gen_mel_weight_matrix(int(num_mel_bins), int(num_spectrogram_bins), float(sample_rate),
                                                float(lower_edge_hertz), float(upper_edge_hertz),
                                                W.ctypes.data_as(ctypes.c_void_p))
    if r != _lib.OK:
        raise _lib.EdisonError(r, "edison_gen_mel_weight_matrix")
    return W


_configured = None


def _is_fast_geometry(frame_len, mel_nbins):
    return frame_len == _lib.FRAME_LEN and mel_nbins == _lib.NUM_MEL


def _prepare(fs, frame_len, mel_nbins, mel_lower_hz, mel_upper_hz, mel_mtx_scale=128):
    """The context, configured for this filterbank when the geometry is the fast kernels' (1024 / 32); other geometries take the
    generality kernel, which builds its tables per call and needs no configuration."""
    global _configured
    if not _is_fast_geometry(frame_len, mel_nbins):
        if not (4 <= int(frame_len) <= 4096 and 1 <= int(mel_nbins) <= 256):   # before anything touches the device
            raise NotImplementedError("the MI355X path implements frame_len 4 .. 4096 and mel_nbins 1 .. 256; got frame_len=%r mel_nbins=%r"
                                      % (frame_len, mel_nbins))
        return default_context()
    ctx = default_context()
    key = (float(fs), float(mel_lower_hz), float(mel_upper_hz), float(mel_mtx_scale))
    default = (16000.0, 80.0, 7600.0, 128.0)
    if key != (_configured or default):
        ctx.configure_mfcc(*key)
        _configured = key
    return ctx


def _generic(ctx, data, frame_count, frame_len, frame_step, variant, mel_nbins, fs, lo, hi, scale=128.0, use_log=False, stages=True):
    """edison_mfcc_generic on a host int16 stream: dict of float64 arrays (fft complex, spectrogram, mel_spectrogram,
    log_mel_spectrogram, mfcc) shaped like the reference's per-frame entries stacked over the frames."""
    x = _as_int16(data).ravel()
    n = int(frame_count)
    if n > 0 and (n - 1) * frame_step + frame_len > x.shape[0]:
        raise ValueError("data too short for %d frames of %d samples" % (n, frame_len))
    fo = frame_len // 2 if variant == _lib.MFCC_A else (frame_len if variant == _lib.MFCC_B else frame_len // 2 + 1)
    fft = np.zeros((max(n, 0), fo, 2), np.float64) if stages else None
    spec = np.zeros((max(n, 0), fo), np.float64) if stages else None
    mel = np.zeros((max(n, 0), mel_nbins), np.float64) if stages else None
    lm = np.zeros((max(n, 0), mel_nbins), np.float64) if stages else None
    mf = np.zeros((max(n, 0), mel_nbins), np.float64)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    v = variant | (_lib.MFCC_USE_LOG if use_log else 0)
    ctx._check(_lib.lib().edison_mfcc_generic(ctx._h, p(x), n, int(frame_len), int(frame_step), v, int(mel_nbins), float(fs), float(lo), float(hi),
                                              float(scale), p(fft), p(spec), p(mel), p(lm), p(mf), 0, None, 1.0))
    return dict(fft=None if fft is None else fft[..., 0] + 1j * fft[..., 1], spectrogram=spec, mel_spectrogram=mel, log_mel_spectrogram=lm, mfcc=mf)


def _frame_count(frame_count, nSamples, frame_len, frame_step):
    if frame_count == 0:
        frame_count = 1 + (nSamples - frame_len) // frame_step  # mfcc_utils.py:154-155
    return int(frame_count)


def _as_int16(data):
    a = np.asarray(data)
    if a.dtype != np.int16:
        if np.issubdtype(a.dtype, np.floating) and not np.all(a == np.round(a)):
            raise ValueError("the GPU path takes 16-bit PCM samples (the reference's 16 kHz/16 bit input)")
        if a.size and (a.min() < -32768 or a.max() > 32767):
            raise ValueError("sample values outside int16")
        a = a.astype(np.int16)
    return np.ascontiguousarray(a)


def batch_mfcc(data, fs, nSamples, frame_len, frame_step, frame_count, fft_len, mel_nbins, mel_lower_hz, mel_upper_hz):
    """Variant A over data[..., samples]; returns [n, frame_count, mel_nbins] (mfcc_utils.py:75-131)."""
    ctx = _prepare(fs, frame_len, mel_nbins, mel_lower_hz, mel_upper_hz)
    frame_count = _frame_count(frame_count, nSamples, frame_len, frame_step)
    data = _as_int16(data)
    print("Running mfcc for %d frames with %d step on %d samples" % (frame_count, frame_step, data.shape[0]))
    data = np.atleast_2d(data)
    if not _is_fast_geometry(frame_len, mel_nbins):
        rows = [_generic(ctx, r, frame_count, frame_len, frame_step, _lib.MFCC_A, mel_nbins, fs, mel_lower_hz, mel_upper_hz, stages=False)["mfcc"] for r in data]
        return np.stack(rows) if rows else np.zeros((0, frame_count, mel_nbins))
    # one C-ABI call, one kernel launch for all rows (the kernel's grouped addressing: row = group, row stride = samples)
    return ctx.mfcc_rows(data, frame_count, frame_step=frame_step, variant=_lib.MFCC_A, n_coef=mel_nbins).astype(np.float64)


def mfcc(data, fs, nSamples, frame_len, frame_step, frame_count, fft_len, mel_nbins, mel_lower_hz, mel_upper_hz,
         dummy=None):
    """Variant A; list of per-frame dicts with the reference's keys (mfcc_utils.py:134-199)."""
    ctx = _prepare(fs, frame_len, mel_nbins, mel_lower_hz, mel_upper_hz)
    frame_count = _frame_count(frame_count, nSamples, frame_len, frame_step)
    if _is_fast_geometry(frame_len, mel_nbins):
        st = ctx.mfcc_stages(_as_int16(data), n_frames=frame_count, frame_step=frame_step, variant=_lib.MFCC_A)
    else:
        st = _generic(ctx, data, frame_count, frame_len, frame_step, _lib.MFCC_A, mel_nbins, fs, mel_lower_hz, mel_upper_hz)
    W = gen_mel_weight_matrix(num_mel_bins=mel_nbins, num_spectrogram_bins=frame_len // 2, sample_rate=fs,
                              lower_edge_hertz=mel_lower_hz, upper_edge_hertz=mel_upper_hz)
    output = []
    for f in range(frame_count):
        frame = {}
        frame['t_start'] = f * frame_step / fs
        frame['t_end'] = (f * frame_step + frame_len) / fs
        frame['fft'] = st['fft'][f, :frame_len // 2].astype(np.complex128)
        frame['spectrogram'] = st['spectrogram'][f, :frame_len // 2].astype(np.float64)
        frame['mel_weight_matrix'] = W
        frame['mel_spectrogram'] = st['mel_spectrogram'][f].astype(np.float64)
        frame['log_mel_spectrogram'] = st['log_mel_spectrogram'][f].astype(np.float64)
        frame['mfcc'] = st['mfcc'][f].astype(np.float64)
        output.append(frame)
    return output


def mfcc_mcu(data, fs, nSamples, frame_len, frame_step, frame_count, fft_len, mel_nbins, mel_lower_hz, mel_upper_hz,
             mel_mtx_scale, use_log=False):
    """Variant B (the features the net was trained on); list of per-frame dicts (mfcc_utils.py:255-323)."""
    ctx = _prepare(fs, frame_len, mel_nbins, mel_lower_hz, mel_upper_hz, mel_mtx_scale)
    frame_count = _frame_count(frame_count, nSamples, frame_len, frame_step)
    generic = not _is_fast_geometry(frame_len, mel_nbins)
    if generic:
        st = _generic(ctx, data, frame_count, frame_len, frame_step, _lib.MFCC_B, mel_nbins, fs, mel_lower_hz, mel_upper_hz, mel_mtx_scale, use_log)
    else:
        st = ctx.mfcc_stages(_as_int16(data), n_frames=frame_count, frame_step=frame_step, variant=_lib.MFCC_B,
                             use_log=use_log)
    W = mel_mtx_scale * gen_mel_weight_matrix(num_mel_bins=mel_nbins, num_spectrogram_bins=frame_len // 2 + 1,
                                              sample_rate=fs, lower_edge_hertz=mel_lower_hz,
                                              upper_edge_hertz=mel_upper_hz)
    half = frame_len // 2
    output = []
    for f in range(frame_count):
        frame = {}
        frame['t_start'] = f * frame_step / fs
        frame['t_end'] = (f * frame_step + frame_len) / fs
        if generic:   # the generality kernel returns the reference's full-length entries as they are (scaled, mirrored)
            frame['fft'] = st['fft'][f]
            frame['spectrogram'] = st['spectrogram'][f]
        else:
            # full-length spectrum of a real signal from its 513 unique bins (conjugate symmetry)
            X = st['fft'][f].astype(np.complex128)
            frame['fft'] = 1.0 / 1024 * np.concatenate([X, np.conj(X[half - 1:0:-1])])
            s = st['spectrogram'][f].astype(np.float64)
            frame['spectrogram'] = np.concatenate([s, s[half - 1:0:-1]])
        frame['mel_weight_matrix'] = W
        frame['mel_spectrogram'] = st['mel_spectrogram'][f].astype(np.float64)
        frame['log_mel_spectrogram'] = st['log_mel_spectrogram'][f].astype(np.float64)
        frame['mfcc'] = st['mfcc'][f].astype(np.float64)
        output.append(frame)
    return output


def dct2Makhoul(x):
    """DCT-II of a 1-D array through one FFT of the same length (Makhoul 1980), as the board tools use it to look at
    the firmware's dct2_q15 stage by stage (mfcc_utils.py:324-343, mfcc_on_mcu.py:370). Returns the same triple:
    (the DCT-II = scipy.fftpack.dct(x, 2), the even/odd reordered input, its FFT)."""
    x = np.asarray(x)
    n = x.shape[0]
    half = (n + 1) // 2
    v = np.empty_like(x)
    v[:half] = x[0::2]                       # even samples ascending ...
    v[half:] = x[1::2][::-1]                 # ... then the odd ones descending
    V = np.fft.fft(v)
    twiddle = 2.0 * np.exp(-1j * np.pi * np.arange(n) / (2.0 * n))
    return (V * twiddle).real, v, V


def mfcc_tf(data, fs, nSamples, frame_len, frame_step, frame_count, fft_len, mel_nbins, mel_lower_hz, mel_upper_hz,
            unused=None):
    """The TensorFlow curve of ``main.py mfcc host`` (mfcc_utils.py:201-253) without TensorFlow: tf.signal.stft with its
    default periodic Hann window, tf.abs, tf.signal.linear_to_mel_weight_matrix over the fft_len/2+1 unique bins,
    ln(x + 1e-6) and tf.signal.mfccs_from_log_mel_spectrograms (DCT-II * rsqrt(2 * mel_nbins)), all on the GPU (variant TF of
    the C-ABI). Same list of per-frame dicts, with the DC bin cut from 'fft', 'spectrogram' and 'mel_weight_matrix' as the
    reference does (:245-249); float32 arithmetic like TensorFlow's. Parity unpinned (see the module docstring)."""
    if fft_len != frame_len:
        raise NotImplementedError("the MI355X path implements fft_len == frame_len (audio/config.py:15-16)")
    ctx = _prepare(fs, frame_len, mel_nbins, mel_lower_hz, mel_upper_hz)
    frame_count = _frame_count(frame_count, nSamples, frame_len, frame_step)
    if _is_fast_geometry(frame_len, mel_nbins):
        st = ctx.mfcc_stages(_as_int16(data), n_frames=frame_count, frame_step=frame_step, variant=_lib.MFCC_TF)
    else:   # any other geometry: the generality kernel (float32 window product, then float64)
        st = _generic(ctx, data, frame_count, frame_len, frame_step, _lib.MFCC_TF, mel_nbins, fs, mel_lower_hz, mel_upper_hz)
        st = {k: (v.astype(np.float32) if k != 'fft' else v) for k, v in st.items()}   # TensorFlow's tensors are float32
    W = gen_mel_weight_matrix(num_mel_bins=mel_nbins, num_spectrogram_bins=fft_len // 2 + 1, sample_rate=fs,
                              lower_edge_hertz=mel_lower_hz, upper_edge_hertz=mel_upper_hz)
    output = []
    for f in range(frame_count):
        frame = {}
        frame['t_start'] = f * frame_step / fs
        frame['t_end'] = (f * frame_step + frame_len) / fs
        frame['fft'] = st['fft'][f, 1:].astype(np.complex64)
        frame['spectrogram'] = st['spectrogram'][f, 1:]
        frame['mel_weight_matrix'] = W[1:].astype(np.float32)
        frame['mel_spectrogram'] = st['mel_spectrogram'][f]
        frame['log_mel_spectrogram'] = st['log_mel_spectrogram'][f]
        frame['mfcc'] = st['mfcc'][f]
        output.append(frame)
    return output
