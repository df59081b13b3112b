"""Continuous keyword spotting on a sample stream -- Python handle on ``edison_stream_*`` (include/edison_hip.h).

Counterpart of the firmware's continuous mode (firmware/src/app.c:288-371, 635-719) and of the host mirror
``kws_on_mcu.hostMicContinuous`` (kws_on_mcu.py:520-600): every new frame yields a fresh inference on the newest
31 MFCC rows. Note the reference's host mirror *prepends* new rows (kws_on_mcu.py:558-567) while the firmware
*appends* them (app.c:706-719); this stream follows the firmware (oldest row first), which is also the order the
network was trained on.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import FRAME_LEN, NET_OUT, EdisonError
from .context import KEYWORDS, default_context


class Stream:
    def __init__(self, ctx=None, hop=FRAME_LEN, chunk_frames=1):
        self.ctx = ctx or default_context()
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        r = self._L.edison_stream_create(self.ctx._h, int(hop), int(chunk_frames), ctypes.byref(h))
        self.ctx._check(r)
        self._h = h
        self.hop, self.chunk = int(hop), int(chunk_frames)

    def close(self):
        if getattr(self, "_h", None):
            self._L.edison_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.ctx._check(self._L.edison_stream_reset(self._h))

    @property
    def frames_seen(self):
        return int(self._L.edison_stream_frames_seen(self._h))

    def push(self, samples):
        """samples: chunk_frames*hop new int16 samples (host). Returns dict(logits, softmax, argmax, keywords)."""
        x = np.ascontiguousarray(samples, dtype=np.int16).ravel()
        if x.shape[0] != self.chunk * self.hop:
            raise ValueError("push needs exactly chunk_frames*hop = %d samples" % (self.chunk * self.hop))
        logits = np.zeros((self.chunk, NET_OUT), np.int8)
        soft = np.zeros((self.chunk, NET_OUT), np.int8)
        am = np.zeros(self.chunk, np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.ctx._check(self._L.edison_stream_push(self._h, p(x), p(logits), p(soft), p(am)))
        return dict(logits=logits, softmax=soft, argmax=am, keywords=[KEYWORDS[i] for i in am])

    def push_t(self, samples, logits=None, softmax=None, argmax=None):
        """Device tensors (torch, int16 / int8 / int32 on the context's GPU); asynchronous on the context's stream."""
        if samples.numel() != self.chunk * self.hop:
            raise ValueError("push needs exactly chunk_frames*hop = %d samples" % (self.chunk * self.hop))
        q = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        self.ctx._check(self._L.edison_stream_push_dev(self._h, q(samples), q(logits), q(softmax), q(argmax)))
