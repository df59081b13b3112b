"""Continuous keyword spotting on a sample stream -- Python handle on ``edison_stream_*`` (include/edison_hip.h).

Counterpart of the firmware's continuous mode (firmware/src/app.c:288-371, 635-719) and of the host mirror
``kws_on_mcu.hostMicContinuous`` (kws_on_mcu.py:520-600): every new frame yields a fresh inference on the newest
31 MFCC rows. Note the reference's host mirror *prepends* new rows (kws_on_mcu.py:558-567) while the firmware
*appends* them (app.c:706-719); this stream follows the firmware (oldest row first), which is also the order the
network was trained on.

``q15=True`` computes the features with the firmware's own Q15 arithmetic (MFCC variant C) instead of the host
float model; ``output_filter=True`` adds the firmware's post-processing of the network output (moving average,
maximum, threshold: app.c:332-356). ``Fsm`` is the firmware's wake-word / location / value state machine
(app.c:727-928) without the LEDs.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import FRAME_LEN, NET_OUT, EdisonError
from .context import KEYWORDS, default_context


class Stream:
    def __init__(self, ctx=None, hop=FRAME_LEN, chunk_frames=1, q15=False, output_filter=False, alpha=0.9,
                 threshold=0.5, graph=None, fsm=False):
        self.ctx = ctx or default_context()
        self._L = _lib.lib()
        o = _lib.StreamOpts()
        self._L.edison_stream_default_opts(ctypes.byref(o))
        o.hop, o.chunk_frames = int(hop), int(chunk_frames)
        o.mfcc_variant = _lib.MFCC_C if q15 else _lib.MFCC_B
        o.filter = 1 if (output_filter or fsm) else 0
        o.fsm = 1 if fsm else 0         # the firmware's state machine as the last GPU stage of every push (edison_stream_fsm)
        o.filter_alpha, o.true_threshold = float(alpha), float(threshold)
        if graph is not None:    # None: the library's default (direct launches unless EDISON_STREAM_GRAPH=1)
            o.launch_mode = 1 if graph else 0
        h = ctypes.c_void_p()
        self.ctx._check(self._L.edison_stream_create_ex(self.ctx._h, ctypes.byref(o), ctypes.byref(h)))
        self._h = h
        self.hop, self.chunk, self.output_filter, self.fsm = int(hop), int(chunk_frames), bool(output_filter or fsm), bool(fsm)
        self._fsm_states = np.zeros(self.chunk, np.int32)
        self._fsm = _lib.Fsm()
        self._bufs = None

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
        """samples: chunk_frames*hop new int16 samples (host). Returns dict(logits, softmax, argmax, keywords) plus,
        with the output filter, filtered [chunk,10] fp32, likely [chunk], spotted [chunk] (-1 = below threshold)."""
        x = samples if (type(samples) is np.ndarray and samples.dtype == np.int16 and samples.ndim == 1 and samples.flags.c_contiguous) \
            else np.ascontiguousarray(samples, dtype=np.int16).ravel()
        if x.shape[0] != self.chunk * self.hop:
            raise ValueError("push needs exactly chunk_frames*hop = %d samples" % (self.chunk * self.hop))
        # the microphone path calls this once per frame: the result buffers and their addresses are made once, the call
        # writes into them, the caller gets copies (10 + 10 + 4 bytes per frame)
        b = self._bufs
        if b is None:
            b = self._bufs = self._make_bufs()
        r = self._push(self._h, x.ctypes.data, b[3], b[4], b[5])
        if r != _lib.OK:
            self.ctx._check(r)
        am = b[2].copy()
        out = dict(logits=b[0].copy(), softmax=b[1].copy(), argmax=am, keywords=[KEYWORDS[i] for i in am])
        if self.output_filter:
            self.ctx._check(self._L.edison_stream_filtered(self._h, b[9], b[10], b[11]))
            out.update(filtered=b[6].copy(), likely=b[7].copy(), spotted=b[8].copy())
        if self.fsm:
            self.ctx._check(self._L.edison_stream_fsm(self._h, ctypes.byref(self._fsm), self._fsm_states.ctypes.data))
            out.update(fsm_states=self._fsm_states.copy(), fsm=self.fsm_snapshot())
        return out

    def fsm_snapshot(self):
        """The state machine as the last edison_stream_fsm call saw it: dict(state, hot_timeout_ms, last_command, commands)."""
        f = self._fsm
        cmd = None if f.last_loc < 0 else (KEYWORDS[f.last_loc], KEYWORDS[f.last_val])
        return dict(state=Fsm.STATES[f.state], hot_timeout_ms=int(f.hot_timeout_ms), last_command=cmd, commands=int(f.commands),
                    raw=(f.state, f.hot_timeout_ms, f.wake_idx, f.loc_idx, f.val_idx, f.last_loc, f.last_val, f.commands))

    def _make_bufs(self):
        c = self.chunk
        lo, so, am = np.zeros((c, NET_OUT), np.int8), np.zeros((c, NET_OUT), np.int8), np.zeros(c, np.int32)
        fl, li, sp = np.zeros((c, NET_OUT), np.float32), np.zeros(c, np.int32), np.zeros(c, np.int32)
        self._push = self._L.edison_stream_push
        return (lo, so, am, lo.ctypes.data, so.ctypes.data, am.ctypes.data, fl, li, sp, fl.ctypes.data, li.ctypes.data, sp.ctypes.data)

    def push_t(self, samples, logits=None, softmax=None, argmax=None, filtered=None, likely=None, spotted=None, n_frames=None):
        """Device tensors (torch, int16 / int8 / int32 / fp32 on the context's GPU); asynchronous on the context's stream.
        n_frames < chunk_frames: a ragged last push (edison_stream_push_n_dev: n_frames * hop samples); every output -- logits, softmax,
        argmax, filtered, likely, spotted -- is [n_frames][..]: the stream copies the entries of THIS push only."""
        n = self.chunk if n_frames is None else int(n_frames)
        if samples.numel() != n * self.hop:
            raise ValueError("push needs exactly n_frames*hop = %d samples" % (n * self.hop))
        q = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        if n_frames is None:
            self.ctx._check(self._L.edison_stream_push_dev(self._h, q(samples), q(logits), q(softmax), q(argmax)))
        else:
            self.ctx._check(self._L.edison_stream_push_n_dev(self._h, q(samples), n, q(logits), q(softmax), q(argmax)))
        if filtered is not None or likely is not None or spotted is not None:
            self.ctx._check(self._L.edison_stream_filtered_dev(self._h, q(filtered), q(likely), q(spotted)))


class Fsm:
    """edisonFSM (app.c:727-928): RESET -> IDLE -> HOT (wake word) -> LOC (location) -> SET (value) -> IDLE."""
    STATES = ("RESET", "IDLE", "HOT", "LOC", "SET")

    def __init__(self, threshold=0.5):
        self._L = _lib.lib()
        self._f = _lib.Fsm()
        self._L.edison_fsm_init(ctypes.byref(self._f))
        self.threshold = float(threshold)

    def step(self, pred_max, pred_idx, dt_us):
        r = self._L.edison_fsm_step(ctypes.byref(self._f), float(pred_max), int(pred_idx), int(dt_us), self.threshold)
        if r < 0:
            raise EdisonError(r, "edison_fsm_step")
        return self.STATES[r]

    @property
    def state(self):
        return self.STATES[self._f.state]

    @property
    def last_command(self):
        """(location, value) keyword strings of the last executed command, or None."""
        if self._f.last_loc < 0:
            return None
        return KEYWORDS[self._f.last_loc], KEYWORDS[self._f.last_val]

    @property
    def commands(self):
        return int(self._f.commands)
