#!/usr/bin/env python3
"""Kernel-only timing of the variant-C (Q15) MFCC kernel, HIP events on the launch stream; A/B work on the box.
usage: bench_q15.py [--frames N] [--reps R] [--lib path/to/libedison_hip.so] [--tag T] [--check]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from edison_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--lib", default=None)
ap.add_argument("--tag", default="")
ap.add_argument("--check", action="store_true", help="compare 4096 frames with the oracle (skip for ablation builds)")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
from edison_amd.context import Context

dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx.use_torch_stream(st)
g = torch.Generator(device=dev)
g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
out = torch.empty((a.frames, 13), dtype=torch.int16, device=dev)
for i in range(3):
    ctx.mfcc_q15_t(bufs[i % 3], a.frames, 1024, 13, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.reps):
    ctx.mfcc_q15_t(bufs[i % 3], a.frames, 1024, 13, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print("%s q15: %.4f ms / %d frames = %.1f Mframes/s = %.0f GB/s (%.2f%% of 8 TB/s)" % (
    a.tag, ms, a.frames, a.frames / ms / 1e3, 2074 * a.frames / ms / 1e6, 2074 * a.frames / ms / 1e6 / 80))
if a.check:
    from oracle import oracle
    n = min(4096, a.frames)
    ctx.mfcc_q15_t(bufs[0], n, 1024, 13, out=out)
    torch.cuda.synchronize()
    ref = oracle.mfcc_q15(bufs[0][:n].cpu().numpy().reshape(-1), n_threads=8)[:, :13]
    print("   oracle check on %d frames:" % n, "bit-exact" if np.array_equal(out[:n].cpu().numpy(), ref) else "MISMATCH")
