"""The ``mfcc host`` script: reads a wav, runs the MFCC variants A ("own"), TF ("tf") and B ("mcu") and prints /
optionally plots the first 13 coefficients per frame -- the counterpart of the reference's
audio/edison/mfcc/mfcc.py:175-221 with its four curves (own, tf, mcu, mcu log: mfcc.py:209-216), the tf curve from the
GPU's variant TF instead of TensorFlow (mfcc_utils.mfcc_tf here), and with plotting optional (``--plot``; the reference
always calls plt.show()).
"""
import os

import numpy as np

from .. import config as cfg
from . import mfcc_utils as mfu

DEFAULT_WAV = 'data/edison_16k_16b.wav'  # mfcc.py:21, relative to the reference's audio/ directory


def run(in_wav):
  import scipy.io.wavfile as wavfile
  in_fs, in_data = wavfile.read(in_wav)
  in_data = np.array(in_data)
  fs = in_fs
  nSamples = len(in_data)
  print("Frame length in seconds = %.3fs" % (cfg.frame_len / fs))
  print("Number of input samples = %d" % (nSamples))
  o_mfcc = mfu.mfcc(in_data, fs, nSamples, cfg.frame_len, cfg.frame_step, cfg.frame_count, cfg.fft_len,
                    cfg.mel_nbins, cfg.mel_lower_hz, cfg.mel_upper_hz)
  o_mfcc_tf = mfu.mfcc_tf(in_data, fs, nSamples, cfg.frame_len, cfg.frame_step, cfg.frame_count, cfg.fft_len,
                          cfg.mel_nbins, cfg.mel_lower_hz, cfg.mel_upper_hz)
  o_mfcc_mcu = mfu.mfcc_mcu(in_data, fs, nSamples, cfg.frame_len, cfg.frame_step, cfg.frame_count, cfg.fft_len,
                            cfg.mel_nbins, cfg.mel_lower_hz, cfg.mel_upper_hz, cfg.mel_mtx_scale)
  first_mfcc, num_mfcc = 0, 13   # mfcc.py:207-208
  cut = lambda o: np.array([x['mfcc'][first_mfcc:first_mfcc + num_mfcc] for x in o], np.float64)
  with np.errstate(invalid='ignore', divide='ignore'):
    mcu_log = np.log(cut(o_mfcc_mcu))   # mfcc.py:215: the logarithm of the coefficients themselves (nan where negative)
  mfccs = [cut(o_mfcc), cut(o_mfcc_tf), cut(o_mfcc_mcu), mcu_log]
  return o_mfcc, o_mfcc_mcu, np.array(mfccs)


CURVES = ['own', 'tf', 'mcu', 'mcu log']   # mfcc.py:218


def main(argv):
  args = [a for a in argv if not a.startswith('--')]
  in_wav = args[0] if args else DEFAULT_WAV
  if not os.path.exists(in_wav):
    print('wav file %s not found (the reference reads %s relative to its audio/ directory)' % (in_wav, DEFAULT_WAV))
    return 1
  o_mfcc, o_mfcc_mcu, mfccs = run(in_wav)
  print(mfccs.shape)
  np.set_printoptions(precision=3, suppress=True, linewidth=160)
  for name, m in zip(CURVES, mfccs):
    print('%s MFCC (frames x 13):' % name)
    print(m)
  if '--plot' in argv:
    import matplotlib.pyplot as plt
    fig, axs = plt.subplots(1, len(CURVES))
    for ax, name, m in zip(axs, CURVES, mfccs):
      ax.pcolor(m.T, cmap='PuBu')
      ax.set_title(name)
    plt.show()
  return 0
