"""ctypes binding of libedison_hip.so -- the same C-ABI (include/edison_hip.h) any other FFI would bind.

There is no Python/NumPy/torch implementation behind these calls and no fallback: if the shared library is
missing the import of the binding raises, and if no gfx950 device is visible ``Context()`` raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDISON_LIB") or os.path.join(_HERE, "csrc", "libedison_hip.so")  # EDISON_LIB: another build
DEFAULT_MODEL = os.path.join(_HERE, "data", "kws_nnom.ednn")

OK = 0
E_ARGUMENT, E_LENGTH, E_SIZE, E_NO_MEMORY, E_MORE_TODO = -1, -2, -3, -7, -8
E_RUNTIME, E_NO_IMPL, E_NO_DEVICE, E_NO_MODEL = -16, -17, -18, -19
MFCC_A, MFCC_B, MFCC_C, MFCC_TF, MFCC_USE_LOG = 0, 1, 2, 3, 0x100

FS, FRAME_LEN, NUM_MEL, NUM_MFCC, UTT_FRAMES, NET_IN, NET_OUT = 16000, 1024, 32, 13, 31, 403, 10
DIST_ID_BYTES = 128
CNN_ACT_BYTES = 10420

c_void_p, c_int, c_int64, c_size_t, c_float, c_double, c_char_p = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float, ctypes.c_double, ctypes.c_char_p)

class StreamOpts(ctypes.Structure):
    """edison_stream_opts"""
    _fields_ = [("hop", c_int), ("chunk_frames", c_int), ("mfcc_variant", c_int), ("filter", c_int),
                ("filter_alpha", c_double), ("true_threshold", c_double), ("launch_mode", c_int), ("fsm", c_int)]


class NetInfo(ctypes.Structure):
    """edison_net_info"""
    _fields_ = [(k, ctypes.c_int32) for k in ("in_h", "in_w", "in_c", "n_out", "n_layers", "acts_bytes", "has_softmax",
                                              "accelerated")]


class NetLayerInfo(ctypes.Structure):
    """edison_net_layer_info_t"""
    _fields_ = [(k, ctypes.c_int32) for k in ("type", "out_h", "out_w", "out_c", "acts_offset", "relu")]


class Fsm(ctypes.Structure):
    """edison_fsm"""
    _fields_ = [("state", c_int), ("hot_timeout_ms", ctypes.c_uint32), ("wake_idx", c_int), ("loc_idx", c_int),
                ("val_idx", c_int), ("last_loc", c_int), ("last_val", c_int), ("commands", ctypes.c_uint32)]


# name -> (restype, argtypes): every symbol include/edison_hip.h declares
SIGNATURES = {
    "edison_init": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "edison_shutdown": (None, [c_void_p]),
    "edison_last_error": (c_char_p, [c_void_p]),
    "edison_set_stream": (c_int, [c_void_p, c_void_p]),
    "edison_reset_stream": (c_int, [c_void_p]),
    "edison_sync": (c_int, [c_void_p]),
    "edison_device_info": (c_int, [c_void_p, c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    "edison_mfcc_configure": (c_int, [c_void_p, c_double, c_double, c_double, c_double]),
    "edison_gen_mel_weight_matrix": (c_int, [c_int, c_int, c_double, c_double, c_double, c_void_p]),
    "edison_model_load": (c_int, [c_void_p, c_char_p]),
    "edison_model_load_mem": (c_int, [c_void_p, c_void_p, c_size_t]),
    "edison_net_get_info": (c_int, [c_void_p, ctypes.POINTER(NetInfo)]),
    "edison_net_layer_info": (c_int, [c_void_p, c_int, ctypes.POINTER(NetLayerInfo)]),
    "edison_net_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "edison_net_layers_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "edison_net_batch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "edison_net_layers": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "edison_net_specialize": (c_int, [c_void_p]),
    "edison_net_specialized": (c_int, [c_void_p]),
    "edison_net_spec_source": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "edison_net_plan_dump": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_size_t), c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "edison_net_plan_layout": (ctypes.c_size_t, [c_int]),
    "edison_dist_available": (c_int, []),
    "edison_dist_unique_id": (c_int, [c_void_p]),
    "edison_dist_init": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "edison_dist_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "edison_dist_shutdown": (c_int, [c_void_p]),
    "edison_dist_shard_range": (c_int, [c_int64, c_int, c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "edison_dist_allgather_logits": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "edison_kws_batch_sharded_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_dist_allgather_logits_total": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "edison_kws_batch_sharded_total_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_dev_alloc": (c_int, [c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    "edison_dev_free": (c_int, [c_void_p, c_void_p]),
    "edison_dev_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "edison_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "edison_mfcc_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float]),
    "edison_mfcc_rows_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float]),
    "edison_mfcc_batches_dev": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float]),
    "edison_mfcc_generic_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_double, c_double, c_double, c_double, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float]),
    "edison_mfcc_generic": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_double, c_double, c_double, c_double, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float]),
    "edison_queues_calibrate": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, ctypes.POINTER(c_double), ctypes.POINTER(c_double), ctypes.POINTER(c_int)]),
    "edison_queues_fork": (c_int, [c_void_p]),
    "edison_queues_join": (c_int, [c_void_p]),
    "edison_mfcc_batch_queue_dev": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float]),
    "edison_mfcc_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float]),
    "edison_mfcc_stages_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "edison_cnn_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "edison_cnn_layers_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "edison_kws_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_mfcc_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float]),
    "edison_mfcc_stages": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "edison_cnn_batch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "edison_cnn_layers": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "edison_kws_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_mfcc_q15_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "edison_mfcc_q15_stages_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_kws_batch_q15_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_mfcc_q15_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "edison_mfcc_q15_stages": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_kws_batch_q15": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_stream_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "edison_stream_destroy": (None, [c_void_p]),
    "edison_stream_reset": (c_int, [c_void_p]),
    "edison_stream_push_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_stream_push_n_dev": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "edison_stream_fsm": (c_int, [c_void_p, c_void_p, c_void_p]),
    "edison_stream_fsm_dev": (c_int, [c_void_p, c_void_p]),
    "edison_postproc": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_double, ctypes.c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_fsm_roles": (None, [c_void_p, c_void_p, c_void_p]),
    "edison_stream_push": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_stream_frames_seen": (c_int64, [c_void_p]),
    "edison_stream_default_opts": (None, [ctypes.POINTER(StreamOpts)]),
    "edison_stream_create_ex": (c_int, [c_void_p, ctypes.POINTER(StreamOpts), ctypes.POINTER(c_void_p)]),
    "edison_stream_filtered": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_stream_filtered_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "edison_fsm_init": (None, [ctypes.POINTER(Fsm)]),
    "edison_fsm_step": (c_int, [ctypes.POINTER(Fsm), c_float, ctypes.c_uint32, ctypes.c_uint32, c_double]),
    # legacy firmware call surface
    "aiInitialize": (c_int, []),
    "aiPrintInfo": (None, []),
    "aiGetInputShape": (None, [ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16)]),
    "aiRunInference": (c_int, [c_void_p, c_void_p]),
    "aiGetKeywordFromIndex": (c_char_p, [ctypes.c_uint32]),
    "aiGetKeywordCount": (ctypes.c_uint32, []),
    "aiNnomInit": (None, []),
    "aiNnomTest": (None, []),
    "aiNnomPrintInfo": (None, []),
    "aiNnomRunInference": (c_int, [c_void_p, c_void_p]),
    "aiNnomPredict": (c_int, [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(c_float)]),
    "aiNnomGetInputBuffer": (c_void_p, []),
    "aiNnomGetOutputBuffer": (c_void_p, []),
    "mfccToNetInput": (None, [c_void_p, ctypes.c_uint16, ctypes.c_uint16, ctypes.c_uint32]),
    "mfccToNetInputPush": (None, [c_void_p, ctypes.c_uint16, ctypes.c_uint16]),
    "audioInit": (None, []),
    "audioCalcMFCCs": (None, [c_void_p, ctypes.POINTER(c_void_p)]),
    "mfcc_create": (c_void_p, [c_int, c_int, c_int, c_int, c_float]),
    "mfcc_delete": (None, [c_void_p]),
    "mfcc_compute": (None, [c_void_p, c_void_p, c_void_p]),
    "create_dct_matrix": (c_void_p, [ctypes.c_int32, ctypes.c_int32]),
    "edison_mfcc_f32_create": (c_void_p, [c_void_p, c_int, c_int, c_int, c_int, c_float]),
    "edison_mfcc_f32_n_out": (c_int, [c_void_p]),
    "edison_mfcc_f32_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "edison_mfcc_f32_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "edison_f32_stream_create": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "edison_f32_stream_destroy": (None, [c_void_p]),
    "edison_f32_stream_reset": (c_int, [c_void_p]),
    "edison_f32_stream_events_seen": (c_int64, [c_void_p]),
    "edison_f32_stream_push_dev": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "edison_f32_stream_push": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "edison_mfcc_frame": (c_int, [c_void_p, c_int, c_void_p]),
    "edison_global_ctx": (c_void_p, []),
}

_lib = None


class EdisonError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libedison_hip error %d: %s" % (code, msg))
        self.code = code


def _share_torch_hip_runtime():
    """A PyTorch-ROCm wheel bundles its own libamdhip64.so.7. Two HIP runtimes in one process cannot both own the
    GPU (whichever comes second reports "No HIP GPUs are available"), so when torch is installed but not imported
    yet, map ITS runtime first: libedison_hip.so's DT_NEEDED then binds to it by soname and a later `import torch`
    finds it already loaded. No torch import, no torch dependency; EDISON_NO_TORCH_HIP=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("EDISON_NO_TORCH_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libedison_hip.so (once). Raises if it has not been built: there is nothing to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s is missing -- build it with `python -m edison_amd.build` (hipcc, gfx950). "
                "edison_amd has no CPU implementation." % LIB_PATH)
        _share_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
