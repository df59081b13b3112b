"""Constants of the keyword-spotting path, under the names the reference's tools import from ``audio/config.py``.

The values are fixed by the reference (audio/config.py:11-48 and firmware/src/ai/nnom/keywords.txt) -- the int8
network was trained on exactly this framing -- so they are data, not tunables. They are kept in one frozen record
and re-exported as module attributes so that ``from config import *`` style callers keep working.
"""
from dataclasses import dataclass, fields


@dataclass(frozen=True)
class KwsConfig:
    # acquisition (config.py:11-13)
    fs: int = 16000                      # sample rate [Hz]
    sample_len_seconds: float = 2.0      # one utterance
    # framing (config.py:19-29): frame = hop = FFT length, no overlap
    frame_length: int = 1024
    # mel filterbank (config.py:14-15)
    num_mel_bins: int = 32
    lower_edge_hertz: float = 80.0
    upper_edge_hertz: float = 7600.0
    mel_mtx_scale: int = 128             # integer scale of the firmware's mel matrix
    mel_twiddle_scale: int = 128
    # cepstral coefficients handed to the network (config.py:23-24)
    first_mfcc: int = 0
    num_mfcc: int = 13
    # mel scale, 1127 * ln(1 + f / 700) (config.py:35-36)
    MEL_HIGH_FREQUENCY_Q: float = 1127.0
    MEL_BREAK_FREQUENCY_HERTZ: float = 700.0
    # NNoM int8 input quantisation (config.py:40-43)
    nnom_net_input_scale: float = 1.0
    nnom_net_input_clip_min: int = -128
    nnom_net_input_clip_max: int = 127

    # ---- derived quantities, same names as the reference module ----
    @property
    def nSamples(self):
        return int(self.sample_len_seconds * self.fs)

    @property
    def n_frames(self):
        return 1 + (self.nSamples - self.frame_length) // self.frame_length


CONFIG = KwsConfig()

# flat re-export: every field, then the aliases the reference module defines for the same quantities
globals().update({f.name: getattr(CONFIG, f.name) for f in fields(CONFIG)})
nSamples = sample_len = CONFIG.nSamples
n_frames = CONFIG.n_frames
sample_rate = CONFIG.fs
mel_nbins = CONFIG.num_mel_bins
mel_lower_hz, mel_upper_hz = CONFIG.lower_edge_hertz, CONFIG.upper_edge_hertz
frame_len = frame_step = sample_size = fft_len = CONFIG.frame_length
num_spectrogram_bins = CONFIG.frame_length // 2 + 1
frame_count = 0  # 0 = "as many frames as fit" in every function that takes it

keywords = ("edison", "cinema", "bedroom", "office", "livingroom", "kitchen", "on", "off", "_cold", "_noise")
