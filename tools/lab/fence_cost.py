#!/usr/bin/env python3
"""What does a kernel boundary on ANOTHER hardware queue cost a running MFCC launch?

One long MFCC launch (--frames, default 1 M frames = ~0.7 ms) runs on stream A; while it runs, N tiny kernels (one element, one
wave: each is a dispatch with the runtime's acquire / release fences around it) are issued on stream B (another priority = another
hardware queue). The long launch's duration against N gives the cost per foreign kernel boundary. usage: fence_cost.py [--frames N]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1 << 20)
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
dev = torch.device("cuda", 0)
A = torch.cuda.Stream(priority=0); B = torch.cuda.Stream(priority=-1)
torch.cuda.set_stream(A)
ctx = Context(0)
ctx.use_torch_stream(A)
g = torch.Generator(device=dev); g.manual_seed(1)
x = (torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16)
out = torch.empty((a.frames, 13), dtype=torch.float32, device=dev)
one = torch.zeros(1, device=dev)

def run(n_ping):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(A)
    B.wait_event(e0)
    ctx.mfcc_t(x, a.frames, 1024, _lib.MFCC_B, 13, out=out)
    e1.record(A)
    with torch.cuda.stream(B):
        p0.record(B)
        for _ in range(n_ping):
            one.add_(1.0)
        p1.record(B)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, p0.elapsed_time(p1) * 1e3

for _ in range(20): run(0)
res = {}
NS = [0, 25, 50, 100, 200]
for r in range(a.rounds):
    for n in (NS if r % 2 == 0 else NS[::-1]):
        res.setdefault(n, []).append(run(n))
base = statistics.median(t for t, _ in res[0])
for n in NS:
    t = statistics.median(t for t, _ in res[n]); p = statistics.median(p for _, p in res[n])
    print("pings %4d: long launch %8.1f us (%+6.1f us, %+5.2f %%), the pings took %7.1f us in all%s" % (
        n, t, t - base, (t / base - 1) * 100, p, "  -> %.2f us of the long launch per foreign kernel" % ((t - base) / n) if n else ""), flush=True)
