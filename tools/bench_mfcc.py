#!/usr/bin/env python3
"""Kernel-only timing of the MFCC / CNN kernels (HIP events on the launch stream), for A/B work on the box.
usage: bench_mfcc.py [--frames N] [--utts N] [--reps R]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--utts", type=int, default=32768)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--tag", default="")
ap.add_argument("--warm", type=int, default=3, help="untimed launches first (2000+ reaches the settled clocks)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.use_torch_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
out = torch.empty((a.frames, 13), dtype=torch.float32, device=dev)
def t_mfcc(variant):
    for i in range(a.warm): ctx.mfcc_t(bufs[i % 3], a.frames, 1024, variant, 13, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.reps): ctx.mfcc_t(bufs[i % 3], a.frames, 1024, variant, 13, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
for v, n in ((_lib.MFCC_B, "B"), (_lib.MFCC_A, "A")):
    ms = t_mfcc(v)
    print("%s mfcc %s: %.4f ms / %d frames = %.1f Mframes/s = %.0f GB/s (%.1f%% of 8 TB/s)" % (
        a.tag, n, ms, a.frames, a.frames / ms / 1e3, 2100 * a.frames / ms / 1e6, 2100 * a.frames / ms / 1e6 / 80))
feat = torch.randint(-128, 128, (a.utts, 403), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
lo = torch.empty((a.utts, 10), dtype=torch.int8, device=dev); am = torch.empty((a.utts,), dtype=torch.int32, device=dev)
for i in range(2): ctx.cnn_t(feat, a.utts, logits=lo, softmax=lo, argmax=am)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5): ctx.cnn_t(feat, a.utts, logits=lo, softmax=lo, argmax=am)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("%s cnn: %.3f ms / %d utt = %.2f Mutt/s = %.1f TOPS(int8 MAC*2)" % (a.tag, ms, a.utts, a.utts / ms / 1e3, a.utts * 784752 * 2 / ms / 1e9))
