"""Network-input features for a whole data set in one launch -- the MFCC leg of the reference's ``load_data``
(audio/edison/train/kws_keras.py:443-468 and its twin in kws_nnom.py): there, every utterance of x_train / x_test / x_val goes
through ``mfcc_mcu`` in a Python loop (tens of thousands of calls, ~50 us per frame on the CPU), the first ``num_mfcc``
coefficients from ``first_mfcc`` on are kept, multiplied by the net-input scale and clipped -- floats, NOT rounded: the int8 rounding
happens on the device (``mfccToNetInput``) and in the inference harness (kws_nnom.py:359-361), training sees the clipped floats --
and a channel axis is appended for the Conv2D input. Here the data set is ONE ``edison_mfcc_rows`` call (variant B, one utterance per
row, the grouped kernel), the rest is the same numpy.

Loading the wav files, the labels and the Keras training itself stay out of scope; this is the caller of the hot path on
its input side."""
import numpy as np

from .. import _lib
from .. import config as cfg
from ..context import default_context
from ..mfcc import mfcc_utils as mfu


def dataset_features(x, fs=cfg.fs, nSamples=cfg.nSamples, frame_length=cfg.frame_length, frame_step=cfg.frame_length, frame_count=0,
                     num_mel_bins=cfg.num_mel_bins, lower_edge_hertz=cfg.lower_edge_hertz, upper_edge_hertz=cfg.upper_edge_hertz,
                     mel_mtx_scale=cfg.mel_mtx_scale, use_mfcc_log=False, first_mfcc=cfg.first_mfcc, num_mfcc=cfg.num_mfcc,
                     net_input_scale=cfg.nnom_net_input_scale, net_input_clip_min=cfg.nnom_net_input_clip_min,
                     net_input_clip_max=cfg.nnom_net_input_clip_max, ctx=None):
    """x: int16 [n_utterances, nSamples] (the reference's x_train ...). Returns float64 [n, frames, num_mfcc, 1] =
    np.expand_dims(np.clip(mfcc[:, :, first_mfcc:first_mfcc + num_mfcc] * net_input_scale, clip_min, clip_max), -1), what
    kws_keras.py:443-468 builds utterance by utterance."""
    x = np.atleast_2d(mfu._as_int16(x))
    if x.shape[1] < nSamples:
        raise ValueError("utterances shorter than nSamples = %d" % nSamples)
    if frame_count == 0:
        frame_count = 1 + (nSamples - frame_length) // frame_step          # mfcc_utils.py:277-278
    n_coef = first_mfcc + num_mfcc
    if mfu._is_fast_geometry(frame_length, num_mel_bins):
        c = mfu._prepare(fs, frame_length, num_mel_bins, lower_edge_hertz, upper_edge_hertz, mel_mtx_scale) if ctx is None else ctx
        m = c.mfcc_rows(np.ascontiguousarray(x[:, :nSamples]), frame_count, frame_step=frame_step, variant=_lib.MFCC_B, n_coef=n_coef,
                        use_log=use_mfcc_log).astype(np.float64)
    else:   # another geometry: the generality kernel, utterance by utterance (float64)
        c = default_context() if ctx is None else ctx
        m = np.stack([mfu._generic(c, r[:nSamples], frame_count, frame_length, frame_step, _lib.MFCC_B, num_mel_bins, fs, lower_edge_hertz,
                                   upper_edge_hertz, mel_mtx_scale, use_mfcc_log, stages=False)["mfcc"][:, :n_coef] for r in x])
    m = np.clip(m[:, :, first_mfcc:n_coef] * net_input_scale, net_input_clip_min, net_input_clip_max)
    return np.expand_dims(m, axis=-1)
