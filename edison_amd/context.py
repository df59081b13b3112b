"""Python handle on one libedison_hip context (one per process per GPU).

Host arrays (numpy) go through the synchronous host-pointer entry points; torch CUDA tensors go through the
``*_dev`` entry points on torch's current stream (PyTorch supplies device memory and streams only -- every
computation is a hand-written HIP kernel behind the C-ABI).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (CNN_ACT_BYTES, FRAME_LEN, MFCC_A, MFCC_B, MFCC_USE_LOG, NET_IN, NET_OUT, NUM_MEL, NUM_MFCC,
                   UTT_FRAMES, EdisonError)

KEYWORDS = ["edison", "cinema", "bedroom", "office", "livingroom", "kitchen", "on", "off", "_cold", "_noise"]


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _t_ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Context:
    def __init__(self, device=0, model_path=_lib.DEFAULT_MODEL):
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        r = self._L.edison_init(int(device), ctypes.byref(h))
        if r != _lib.OK:
            raise EdisonError(r, (self._L.edison_last_error(None) or b"").decode())
        self._h = h
        self.device = int(device)
        if model_path is not None:
            self.load_model(model_path)

    # ------------------------------------------------------------------ plumbing
    def _check(self, r):
        if r != _lib.OK:
            raise EdisonError(r, (self._L.edison_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.edison_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_model(self, path):
        self._check(self._L.edison_model_load(self._h, str(path).encode()))

    def load_weights_h(self, path):
        """Load an NNoM-generated model header directly (the reference's weights.h, or one of a retrained / other graph)."""
        from . import nnom_import
        with open(path, "r") as f:
            shape, layers = nnom_import.parse_weights_h(f.read())
        self.load_model_bytes(nnom_import.build_blob(shape, layers))

    def load_model_bytes(self, blob):
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        self._check(self._L.edison_model_load_mem(self._h, ctypes.cast(buf, ctypes.c_void_p), len(blob)))

    def configure_mfcc(self, sample_rate=16000, lower_edge_hertz=80.0, upper_edge_hertz=7600.0, mel_mtx_scale=128):
        self._check(self._L.edison_mfcc_configure(self._h, float(sample_rate), float(lower_edge_hertz),
                                                  float(upper_edge_hertz), float(mel_mtx_scale)))

    def device_info(self):
        name = ctypes.create_string_buffer(128)
        ncu, hbm = ctypes.c_int(), ctypes.c_int64()
        self._check(self._L.edison_device_info(self._h, name, 128, ctypes.byref(ncu), ctypes.byref(hbm)))
        return dict(name=name.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value)

    def use_torch_stream(self, stream=None):
        """Enqueue on a torch CUDA stream (default: torch's current stream of this device). Call it before the
        *_t entry points so the kernels are ordered with the torch ops that produce / consume the tensors."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._check(self._L.edison_set_stream(self._h, ctypes.c_void_p(s.cuda_stream)))

    def use_own_stream(self):
        self._check(self._L.edison_reset_stream(self._h))

    def sync(self):
        self._check(self._L.edison_sync(self._h))

    # ------------------------------------------------------------------ host (numpy) entry points
    def mfcc(self, audio, n_frames=None, frame_step=FRAME_LEN, variant=MFCC_B, n_coef=NUM_MEL, use_log=False,
             want_feat=False, feat_scale=1.0):
        """audio: 1-D int16 stream -> fp32 [n_frames, n_coef] (and int8 net input if want_feat)."""
        x = np.ascontiguousarray(audio, dtype=np.int16).ravel()
        if n_frames is None:
            n_frames = 1 + (x.shape[0] - FRAME_LEN) // frame_step if x.shape[0] >= FRAME_LEN else 0
        if n_frames > 0 and (n_frames - 1) * frame_step + FRAME_LEN > x.shape[0]:
            raise ValueError("audio too short for %d frames" % n_frames)
        out = np.zeros((max(n_frames, 0), n_coef), np.float32)
        feat = np.zeros((max(n_frames, 0), n_coef), np.int8) if want_feat else None
        v = variant | (MFCC_USE_LOG if use_log else 0)
        self._check(self._L.edison_mfcc_batch(self._h, _np_ptr(x), n_frames, frame_step, v, n_coef, _np_ptr(out),
                                              _np_ptr(feat), float(feat_scale)))
        return (out, feat) if want_feat else out

    def mfcc_rows(self, data, frames_per_row, frame_step=FRAME_LEN, variant=MFCC_B, n_coef=NUM_MEL, use_log=False):
        """data: int16 [n_rows, samples] -> fp32 [n_rows, frames_per_row, n_coef] in ONE launch (the reference's
        batch_mfcc shape, mfcc_utils.py:75-131)."""
        x = np.ascontiguousarray(data, dtype=np.int16)
        if x.ndim != 2:
            raise ValueError("data must be [n_rows, samples]")
        n_rows, ns = x.shape
        if frames_per_row > 0 and (frames_per_row - 1) * frame_step + FRAME_LEN > ns:
            raise ValueError("rows too short for %d frames" % frames_per_row)
        out = np.zeros((n_rows, max(frames_per_row, 0), n_coef), np.float32)
        v = variant | (MFCC_USE_LOG if use_log else 0)
        self._check(self._L.edison_mfcc_rows(self._h, _np_ptr(x), n_rows, ns, int(frames_per_row), int(frame_step), v,
                                             int(n_coef), _np_ptr(out), None, 1.0))
        return out

    def mfcc_stages(self, audio, n_frames=None, frame_step=FRAME_LEN, variant=MFCC_B, use_log=False):
        x = np.ascontiguousarray(audio, dtype=np.int16).ravel()
        if n_frames is None:
            n_frames = 1 + (x.shape[0] - FRAME_LEN) // frame_step if x.shape[0] >= FRAME_LEN else 0
        if n_frames > 0 and (n_frames - 1) * frame_step + FRAME_LEN > x.shape[0]:
            raise ValueError("audio too short for %d frames" % n_frames)
        n = max(n_frames, 0)
        fft = np.zeros((n, 513, 2), np.float32)
        spec = np.zeros((n, 513), np.float32)
        mel = np.zeros((n, 32), np.float32)
        logmel = np.zeros((n, 32), np.float32)
        mfcc = np.zeros((n, 32), np.float32)
        v = variant | (MFCC_USE_LOG if use_log else 0)
        self._check(self._L.edison_mfcc_stages(self._h, _np_ptr(x), n_frames, frame_step, v, _np_ptr(fft),
                                               _np_ptr(spec), _np_ptr(mel), _np_ptr(logmel), _np_ptr(mfcc)))
        return dict(fft=fft[..., 0] + 1j * fft[..., 1], spectrogram=spec, mel_spectrogram=mel,
                    log_mel_spectrogram=logmel, mfcc=mfcc)

    def cnn(self, feat):
        f = np.ascontiguousarray(feat, dtype=np.int8).reshape(-1, NET_IN)
        n = f.shape[0]
        logits = np.zeros((n, NET_OUT), np.int8)
        soft = np.zeros((n, NET_OUT), np.int8)
        am = np.zeros(n, np.int32)
        self._check(self._L.edison_cnn_batch(self._h, _np_ptr(f), n, _np_ptr(logits), _np_ptr(soft), _np_ptr(am)))
        return dict(logits=logits, softmax=soft, argmax=am)

    def cnn_layers(self, feat):
        f = np.ascontiguousarray(feat, dtype=np.int8).reshape(-1, NET_IN)
        n = f.shape[0]
        acts = np.zeros((n, CNN_ACT_BYTES), np.int8)
        self._check(self._L.edison_cnn_layers(self._h, _np_ptr(f), n, _np_ptr(acts)))
        names = [("conv1", 3888), ("pool1", 1872), ("conv2", 2464), ("pool2", 1120), ("conv3", 960), ("conv4", 96),
                 ("dense", 10), ("softmax", 10)]
        out, off = {}, 0
        for k, sz in names:
            out[k] = acts[:, off:off + sz]
            off += sz
        return out

    # ---- any NNoM graph (edison_net_*): shapes come from the loaded model
    def net_info(self):
        """dict of edison_net_info plus `layers`: one dict per compute layer (type, out_h, out_w, out_c, acts_offset, relu)."""
        info = _lib.NetInfo()
        self._check(self._L.edison_net_get_info(self._h, ctypes.byref(info)))
        d = {k: getattr(info, k) for k, _ in _lib.NetInfo._fields_}
        d["layers"] = []
        for i in range(info.n_layers):
            li = _lib.NetLayerInfo()
            self._check(self._L.edison_net_layer_info(self._h, i, ctypes.byref(li)))
            d["layers"].append({k: getattr(li, k) for k, _ in _lib.NetLayerInfo._fields_})
        return d

    def net_specialize(self):
        """Compile or fetch from the cache the loaded graph's OWN matrix-core kernel: edison_net_specialize. Returns 1 (compiled
        now by a hipcc child process), 2 (from the cache) or 3 (compiled now by hipRTC in this process); raises EdisonError
        (NO_IMPL) when the graph has no matrix-core plan or no compiler is installed -- the graph then stays on the general kernel."""
        self._check(self._L.edison_net_specialize(self._h))
        return self.net_specialized()

    def net_specialized(self):
        """0: general kernel; the graph's own kernel: 1 compiled now by hipcc, 2 from the on-disk cache, 3 compiled now by hipRTC."""
        return int(self._L.edison_net_specialized(self._h))

    def net(self, x):
        """model_run + first-maximum argmax for n inputs [n][in_h*in_w*in_c] int8 (nnom.c:975-1040, nnom_utils.c:275-284)."""
        info = self.net_info()
        f = np.ascontiguousarray(x, dtype=np.int8).reshape(-1, info["in_h"] * info["in_w"] * info["in_c"])
        n = f.shape[0]
        logits = np.zeros((n, info["n_out"]), np.int8)
        soft = np.zeros((n, info["n_out"]), np.int8) if info["has_softmax"] else None
        am = np.zeros(n, np.int32)
        self._check(self._L.edison_net_batch(self._h, _np_ptr(f), n, _np_ptr(logits), _np_ptr(soft) if soft is not None else None,
                                             _np_ptr(am)))
        return dict(logits=logits, softmax=soft, argmax=am)

    def net_layers(self, x):
        """Every compute layer's output, back to back per input: what a model_set_callback hook sees (nnom.c:1043)."""
        info = self.net_info()
        f = np.ascontiguousarray(x, dtype=np.int8).reshape(-1, info["in_h"] * info["in_w"] * info["in_c"])
        acts = np.zeros((f.shape[0], info["acts_bytes"]), np.int8)
        self._check(self._L.edison_net_layers(self._h, _np_ptr(f), f.shape[0], _np_ptr(acts)))
        return acts

    def _frames(self, x, n_frames, frame_step):
        if n_frames is None:
            n_frames = 1 + (x.shape[0] - FRAME_LEN) // frame_step if x.shape[0] >= FRAME_LEN else 0
        if n_frames > 0 and (n_frames - 1) * frame_step + FRAME_LEN > x.shape[0]:
            raise ValueError("audio too short for %d frames" % n_frames)
        return max(n_frames, 0)

    def mfcc_q15(self, audio, n_frames=None, frame_step=FRAME_LEN, n_coef=NUM_MEL, want_feat=False):
        """Variant C, the firmware's audioCalcMFCCs: int16 [n_frames, n_coef] (and the NNoM int8 net input)."""
        x = np.ascontiguousarray(audio, dtype=np.int16).ravel()
        n = self._frames(x, n_frames, frame_step)
        out = np.zeros((n, n_coef), np.int16)
        feat = np.zeros((n, n_coef), np.int8) if want_feat else None
        self._check(self._L.edison_mfcc_q15_batch(self._h, _np_ptr(x), n, frame_step, n_coef, _np_ptr(out), _np_ptr(feat)))
        return (out, feat) if want_feat else out

    def mfcc_q15_stages(self, audio, n_frames=None, frame_step=FRAME_LEN):
        """What the firmware's audioDumpToHost sends: FFT, spectrum, mel spectrum and DCT output, all int16."""
        x = np.ascontiguousarray(audio, dtype=np.int16).ravel()
        n = self._frames(x, n_frames, frame_step)
        fft = np.zeros((n, 513, 2), np.int16)
        spec = np.zeros((n, 513), np.int16)
        mel = np.zeros((n, 32), np.int16)
        mfcc = np.zeros((n, 32), np.int16)
        self._check(self._L.edison_mfcc_q15_stages(self._h, _np_ptr(x), n, frame_step, _np_ptr(fft), _np_ptr(spec),
                                                   _np_ptr(mel), _np_ptr(mfcc)))
        return dict(fft=fft, spectrogram=spec, mel_spectrogram=mel, mfcc=mfcc)

    def kws(self, audio, n_utt=None, utt_stride=32000, q15=False):
        """audio: int16, utterance u starts at u*utt_stride and uses 31*1024 samples. q15: the firmware's own
        features (variant C) instead of the host float model (variant B)."""
        x = np.ascontiguousarray(audio, dtype=np.int16).ravel()
        used = UTT_FRAMES * FRAME_LEN
        if n_utt is None:
            n_utt = 0 if x.shape[0] < used else 1 + (x.shape[0] - used) // utt_stride
        if n_utt > 0 and (n_utt - 1) * utt_stride + used > x.shape[0]:
            raise ValueError("audio too short for %d utterances" % n_utt)
        feat = np.zeros((n_utt, NET_IN), np.int8)
        logits = np.zeros((n_utt, NET_OUT), np.int8)
        soft = np.zeros((n_utt, NET_OUT), np.int8)
        am = np.zeros(n_utt, np.int32)
        fn = self._L.edison_kws_batch_q15 if q15 else self._L.edison_kws_batch
        self._check(fn(self._h, _np_ptr(x), n_utt, utt_stride, _np_ptr(feat), _np_ptr(logits), _np_ptr(soft), _np_ptr(am)))
        return dict(feat=feat, logits=logits, softmax=soft, argmax=am)

    # ------------------------------------------------------------------ device (torch tensor) entry points
    def mfcc_t(self, audio, n_frames, frame_step=FRAME_LEN, variant=MFCC_B, n_coef=NUM_MFCC, out=None, feat=None,
               feat_scale=1.0, use_log=False):
        """audio: int16 CUDA tensor; out: fp32 [n_frames, n_coef] CUDA tensor or None; feat: int8 or None."""
        v = variant | (MFCC_USE_LOG if use_log else 0)
        self._check(self._L.edison_mfcc_batch_dev(self._h, _t_ptr(audio), int(n_frames), int(frame_step), v, int(n_coef),
                                                  _t_ptr(out), _t_ptr(feat), float(feat_scale)))

    def mfcc_rows_t(self, audio, n_rows, row_stride, frames_per_row, frame_step=FRAME_LEN, variant=MFCC_B, n_coef=NUM_MFCC, out=None,
                    feat=None, feat_scale=1.0):
        """edison_mfcc_rows_dev: `n_rows` rows (utterances of batch_mfcc, or whole BATCHES that sit row_stride samples apart) of
        frames_per_row frames each, in ONE launch; out: fp32 [n_rows * frames_per_row, n_coef]."""
        self._check(self._L.edison_mfcc_rows_dev(self._h, _t_ptr(audio), int(n_rows), int(row_stride), int(frames_per_row), int(frame_step),
                                                 variant, int(n_coef), _t_ptr(out), _t_ptr(feat), float(feat_scale)))

    def mfcc_batches_t(self, audios, n_frames_each, frame_step=FRAME_LEN, variant=MFCC_B, n_coef=NUM_MFCC, outs=None, feats=None,
                       feat_scale=1.0, use_log=False):
        """edison_mfcc_batches_dev: a LIST of independent batches (int16 CUDA tensors at any addresses) with their own outputs (lists of
        fp32 / int8 tensors [n_frames_each, n_coef], or None), ONE launch per 16 batches."""
        n = len(audios)
        arr = (ctypes.c_void_p * n)
        a = arr(*[t.data_ptr() for t in audios])
        o = arr(*[t.data_ptr() for t in outs]) if outs is not None else None
        f = arr(*[t.data_ptr() for t in feats]) if feats is not None else None
        v = variant | (MFCC_USE_LOG if use_log else 0)
        self._check(self._L.edison_mfcc_batches_dev(self._h, n, a, int(n_frames_each), int(frame_step), v, int(n_coef), o, f, float(feat_scale)))

    def queues_calibrate(self, audio, n_frames, frame_step=FRAME_LEN, variant=MFCC_B):
        """edison_queues_calibrate on a representative batch (int16 CUDA tensor): dict(serial_us, best_us, pair) -- pair None when no pair
        of streams beat the serial sequence by 1 % (the queue calls then run serially)."""
        su, bu, pk = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._check(self._L.edison_queues_calibrate(self._h, _t_ptr(audio), int(n_frames), int(frame_step), int(variant), ctypes.byref(su), ctypes.byref(bu),
                                                    ctypes.byref(pk)))
        return dict(serial_us=su.value, best_us=bu.value, pair=(pk.value // 10, pk.value % 10) if pk.value else None)

    def queues_fork(self):
        """Both of the context's two queues start behind the context's stream (edison_queues_fork)."""
        self._check(self._L.edison_queues_fork(self._h))

    def queues_join(self):
        self._check(self._L.edison_queues_join(self._h))

    def mfcc_queue_call(self, queue, audio, n_frames, frame_step=FRAME_LEN, variant=MFCC_B, n_coef=NUM_MFCC, out=None, feat=None, feat_scale=1.0):
        """A prepared edison_mfcc_batch_queue_dev call: returns a function of no arguments that enqueues THIS batch on `queue` (0 / 1)
        between queues_fork() and queues_join(). The ctypes arguments are converted once: a Python host spends ~10 us per call
        this way instead of ~25, which is what keeping two queues fed takes (the GPU needs ~45 us per 65 536-frame batch)."""
        fn = self._L.edison_mfcc_batch_queue_dev
        args = (self._h, ctypes.c_int(int(queue)), _t_ptr(audio), ctypes.c_int64(int(n_frames)), ctypes.c_int64(int(frame_step)), ctypes.c_int(int(variant)),
                ctypes.c_int(int(n_coef)), _t_ptr(out), _t_ptr(feat), ctypes.c_float(float(feat_scale)))
        keep = (audio, out, feat)

        def call(_keep=keep):
            r = fn(*args)
            if r != _lib.OK:
                self._check(r)
        return call

    def cnn_t(self, feat, n_utt, logits=None, softmax=None, argmax=None):
        self._check(self._L.edison_cnn_batch_dev(self._h, _t_ptr(feat), int(n_utt), _t_ptr(logits), _t_ptr(softmax),
                                                 _t_ptr(argmax)))

    def net_t(self, x, n, logits=None, softmax=None, argmax=None):
        """Any loaded graph on device tensors (edison_net_batch_dev)."""
        self._check(self._L.edison_net_batch_dev(self._h, _t_ptr(x), int(n), _t_ptr(logits), _t_ptr(softmax), _t_ptr(argmax)))

    def kws_t(self, audio, n_utt, utt_stride, feat=None, logits=None, softmax=None, argmax=None, q15=False):
        fn = self._L.edison_kws_batch_q15_dev if q15 else self._L.edison_kws_batch_dev
        self._check(fn(self._h, _t_ptr(audio), int(n_utt), int(utt_stride), _t_ptr(feat), _t_ptr(logits),
                       _t_ptr(softmax), _t_ptr(argmax)))

    # ------------------------------------------------------------------ multi-GPU (edison_dist_*: RCCL behind the C-ABI)
    def dist_init(self, id_bytes, rank, world_size):
        buf = ctypes.create_string_buffer(bytes(id_bytes), _lib.DIST_ID_BYTES)
        self._check(self._L.edison_dist_init(self._h, ctypes.cast(buf, ctypes.c_void_p), int(rank), int(world_size)))

    def dist_info(self):
        r, w = ctypes.c_int(), ctypes.c_int()
        self._check(self._L.edison_dist_info(self._h, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def dist_shutdown(self):
        self._check(self._L.edison_dist_shutdown(self._h))

    def allgather_logits_t(self, local_logits, n_local, out):
        self._check(self._L.edison_dist_allgather_logits(self._h, _t_ptr(local_logits), int(n_local), _t_ptr(out)))

    def allgather_logits_total_t(self, local_logits, n_total, out):
        """Shards cut by shard_range(n_total, rank, world): any n_total, one padded ncclAllGather inside."""
        self._check(self._L.edison_dist_allgather_logits_total(self._h, _t_ptr(local_logits), int(n_total), _t_ptr(out)))

    def kws_sharded_total_t(self, audio, n_total, utt_stride, logits_all, feat=None, logits=None, softmax=None, argmax=None):
        self._check(self._L.edison_kws_batch_sharded_total_dev(self._h, _t_ptr(audio), int(n_total), int(utt_stride), _t_ptr(feat),
                                                               _t_ptr(logits), _t_ptr(softmax), _t_ptr(argmax), _t_ptr(logits_all)))

    def kws_sharded_t(self, audio, n_local, utt_stride, logits_all, feat=None, logits=None, softmax=None, argmax=None):
        self._check(self._L.edison_kws_batch_sharded_dev(self._h, _t_ptr(audio), int(n_local), int(utt_stride), _t_ptr(feat),
                                                         _t_ptr(logits), _t_ptr(softmax), _t_ptr(argmax), _t_ptr(logits_all)))

    def mfcc_q15_t(self, audio, n_frames, frame_step=FRAME_LEN, n_coef=NUM_MFCC, out=None, feat=None):
        """audio: int16 CUDA tensor; out: int16 [n_frames, n_coef] CUDA tensor or None; feat: int8 or None."""
        self._check(self._L.edison_mfcc_q15_batch_dev(self._h, _t_ptr(audio), int(n_frames), int(frame_step), int(n_coef),
                                                      _t_ptr(out), _t_ptr(feat)))


_default = None


def postproc(ctx, softmax, alpha=0.9, threshold=0.5, dt_us=64000, state=None, fsm=None):
    """The firmware's post-processing chain (app.c:332-371) on consecutive int8 softmax rows, one GPU stage (edison_postproc):
    returns dict(filtered [n,10] f32, likely [n], spotted [n], state [10] f32, fsm_states [n], fsm (the _lib.Fsm struct)).
    state / fsm: carried in from an earlier call (None: zeros / a machine in RESET)."""
    import ctypes
    L = _lib.lib()
    s = np.ascontiguousarray(softmax, dtype=np.int8).reshape(-1, 10)
    n = s.shape[0]
    st = np.zeros(10, np.float32) if state is None else np.array(state, dtype=np.float32)
    if fsm is None:
        fsm = _lib.Fsm()
        L.edison_fsm_init(ctypes.byref(fsm))
    filt = np.zeros((n, 10), np.float32)
    likely, spotted, states = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    ctx._check(L.edison_postproc(ctx._h, s.ctypes.data, n, float(alpha), float(threshold), int(dt_us), st.ctypes.data, ctypes.byref(fsm),
                                 filt.ctypes.data, likely.ctypes.data, spotted.ctypes.data, states.ctypes.data))
    return dict(filtered=filt, likely=likely, spotted=spotted, state=st, fsm_states=states, fsm=fsm)


def default_context():
    """Process-wide context on cuda:LOCAL_RANK (or 0); created on first use."""
    global _default
    if _default is None:
        import os
        _default = Context(int(os.environ.get("EDISON_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default


def net_spec_source(blob):
    """The constants edison_net_specialize() puts in front of the general kernel's source for this .ednn blob (host only, no
    GPU): edison_net_spec_source. Raises EdisonError (NO_IMPL) for a graph without a matrix-core plan."""
    L = _lib.lib()
    buf = ctypes.create_string_buffer(bytes(blob), len(blob))
    need = ctypes.c_size_t(0)
    r = L.edison_net_spec_source(ctypes.cast(buf, ctypes.c_void_p), len(blob), None, 0, ctypes.byref(need))
    if r != _lib.E_SIZE:
        raise _lib.EdisonError(r, "edison_net_spec_source")
    out = ctypes.create_string_buffer(need.value)
    r = L.edison_net_spec_source(ctypes.cast(buf, ctypes.c_void_p), len(blob), ctypes.cast(out, ctypes.c_void_p), need.value, ctypes.byref(need))
    if r != 0:
        raise _lib.EdisonError(r, "edison_net_spec_source")
    return out.value.decode()
