#!/usr/bin/env python3
"""The HOST-pointer entry points (what a drop-in caller of the reference's Python functions uses): wall time of edison_mfcc_batch and
edison_kws_batch by batch size, against the device-pointer kernel time. usage (GPU box): tools/lab/host_path.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from edison_amd import _lib
from edison_amd.context import Context
ctx = Context(0)
rng = np.random.default_rng(1)
for n in (1, 31, 1024, 16384, 65536, 262144):
    x = np.clip(rng.normal(0, 3000, n * 1024), -32768, 32767).astype(np.int16)
    ctx.mfcc(x, n_frames=n, n_coef=13)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.mfcc(x, n_frames=n, n_coef=13); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print("edison_mfcc_batch (host pointers) %7d frames: %9.3f ms  = %8.2f M frames/s  (%.2f GB/s of samples)" % (n, t * 1e3, n / t / 1e6, n * 2048 / t / 1e9), flush=True)
for nu in (1, 64, 4096, 16384):
    a = np.clip(rng.normal(0, 3000, nu * 32000), -32768, 32767).astype(np.int16)
    ctx.kws(a, n_utt=nu)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.kws(a, n_utt=nu); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print("edison_kws_batch  (host pointers) %7d utterances: %9.3f ms  = %8.3f M inferences/s (%.2f GB/s of samples)" % (nu, t * 1e3, nu / t / 1e6, nu * 64000 / t / 1e9), flush=True)
