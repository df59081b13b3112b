#!/usr/bin/env python3
"""Where a one-frame streaming push spends its time: back-to-back launches of each part on one stream (time per launch
= kernel + launch gap), and host-timed launch + synchronize of the smallest possible graph-free call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.use_torch_stream(st)
audio = (torch.randn(1024, device=dev) * 3000).to(torch.int16)
feat8 = torch.zeros((1, 13), dtype=torch.int8, device=dev)
out = torch.empty((1, 13), dtype=torch.float32, device=dev)
featw = torch.randint(-100, 100, (1, 403), dtype=torch.int32, device=dev).to(torch.int8)
lo = torch.empty((1, 10), dtype=torch.int8, device=dev); so = torch.empty_like(lo); am = torch.empty((1,), dtype=torch.int32, device=dev)
def rep(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def host(fn, n=1000):
    for _ in range(50): fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort(); return ts[len(ts) // 2]
mf = lambda: ctx.mfcc_t(audio, 1, 1024, _lib.MFCC_B, 13, out=out)
cn = lambda: ctx.cnn_t(featw, 1, logits=lo, softmax=so, argmax=am)
print("MFCC, 1 frame: %.1f us per back-to-back launch; host launch + sync %.1f us" % (rep(mf), host(mf)))
print("CNN, 1 utterance: %.1f us per back-to-back launch; host launch + sync %.1f us" % (rep(cn), host(cn)))
x = torch.zeros(4, device=dev)
print("torch x.add_(1) (an empty-ish kernel): %.1f us back-to-back; host launch + sync %.1f us" % (rep(lambda: x.add_(1)), host(lambda: x.add_(1))))
