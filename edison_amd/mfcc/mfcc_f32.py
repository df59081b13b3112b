"""MFCC variant D -- the firmware's float32 ML-KWS feature extractor (firmware/src/audio/mfcc.c), on the GPU.

Python handle on ``mfcc_create`` / ``mfcc_compute`` / ``mfcc_delete`` (firmware/src/audio/mfcc.h:64-67) and their
batched forms in ``include/edison_hip.h``. Defaults are the call of the firmware's NNoM example (app.c:540):
13 features of which the first is dropped, 512-sample frames, dec_bits 8, pre-emphasis 0.97; that example advances by
256 samples per frame (app.c:583).
"""
import ctypes

import numpy as np

from .. import _lib
from ..context import default_context


class MfccF32:
    def __init__(self, num_mfcc_features=13, feature_offset=1, frame_len=512, mfcc_dec_bits=8, preemph=0.97, ctx=None):
        self.ctx = ctx or default_context()
        self._L = _lib.lib()
        self._h = self._L.edison_mfcc_f32_create(self.ctx._h, int(num_mfcc_features), int(feature_offset), int(frame_len),
                                                 int(mfcc_dec_bits), float(preemph))
        if not self._h:
            raise _lib.EdisonError(_lib.E_ARGUMENT, self._L.edison_last_error(self.ctx._h).decode())
        self.frame_len = int(frame_len)
        self.n_out = int(self._L.edison_mfcc_f32_n_out(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.mfcc_delete(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute(self, audio, n_frames=None, frame_step=None, want_float=False):
        """audio: 1-D int16 stream -> int8 [n_frames, n_out] (plus the pre-rounding floats and the log-mel energies)."""
        x = np.ascontiguousarray(audio, dtype=np.int16).ravel()
        step = self.frame_len if frame_step is None else int(frame_step)
        if n_frames is None:
            n_frames = 1 + (x.shape[0] - self.frame_len) // step if x.shape[0] >= self.frame_len else 0
        n = max(int(n_frames), 0)
        if n and (n - 1) * step + self.frame_len > x.shape[0]:
            raise ValueError("audio too short for %d frames" % n)
        out = np.zeros((n, self.n_out), np.int8)
        f32 = np.zeros((n, self.n_out), np.float32) if want_float else None
        lm = np.zeros((n, 26), np.float32) if want_float else None
        p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        self.ctx._check(self._L.edison_mfcc_f32_batch(self._h, p(x), n, step, p(out), p(f32), p(lm)))
        return (out, f32, lm) if want_float else out

    def compute_t(self, audio, n_frames, frame_step, out, out_f32=None, logmel=None):
        """Device tensors (torch): int16 audio, int8 out [n, n_out]; asynchronous on the context's stream."""
        q = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        self.ctx._check(self._L.edison_mfcc_f32_batch_dev(self._h, q(audio), int(n_frames), int(frame_step), q(out),
                                                          q(out_f32), q(logmel)))


class NnomKwsFrontEnd:
    """The audio front end of the firmware's NNoM keyword-spotting example (appNnomKwsRun, app.c:545-623) on the GPU:
    push 512 new samples per event, get the 63 x 12 int8 network input (oldest feature row first) after every event.
    ``dma_to_int16`` is the firmware's conversion of the microphone's 32-bit DMA words (app.c:572-575)."""

    AUDIO_FRAME_LEN, MFCC_LEN = 512, 63            # app.c:497,499

    def __init__(self, ctx=None, window_rows=63, max_events=64, **mfcc_params):
        self.mfcc = MfccF32(ctx=ctx, **mfcc_params)  # defaults = mfcc_create(13, 1, 512, 8, 0.97f), app.c:540
        self.ctx, self._L = self.mfcc.ctx, self.mfcc._L
        h = ctypes.c_void_p()
        self.ctx._check(self._L.edison_f32_stream_create(self.ctx._h, self.mfcc._h, int(window_rows), int(max_events), ctypes.byref(h)))
        self._h, self.rows, self.max_events = h, int(window_rows), int(max_events)

    @staticmethod
    def dma_to_int16(raw32):
        return np.clip(np.asarray(raw32, dtype=np.int32) >> 8, -32768, 32767).astype(np.int16)

    def push(self, samples):
        """samples: k * 512 new int16 samples -> int8 [k, window_rows, n_out]: mfcc_features_seq after each event."""
        x = np.ascontiguousarray(samples, dtype=np.int16).ravel()
        if x.size % self.AUDIO_FRAME_LEN:
            raise ValueError("an audio event is %d samples" % self.AUDIO_FRAME_LEN)
        k = x.size // self.AUDIO_FRAME_LEN
        out = np.zeros((k, self.rows, self.mfcc.n_out), np.int8)
        for lo in range(0, k, self.max_events):
            n = min(self.max_events, k - lo)
            self.ctx._check(self._L.edison_f32_stream_push(self._h, x[lo * 512:].ctypes.data_as(ctypes.c_void_p), n,
                                                           out[lo:].ctypes.data_as(ctypes.c_void_p)))
        return out

    @property
    def events_seen(self):
        return int(self._L.edison_f32_stream_events_seen(self._h))

    def reset(self):
        self.ctx._check(self._L.edison_f32_stream_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.edison_f32_stream_destroy(self._h)
            self._h = None
        self.mfcc.close()
