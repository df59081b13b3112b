#!/usr/bin/env python3
"""Does a trickle of tiny kernels on ANOTHER hardware queue change the serial 65 536-frame launch sequence? (r05: a 1 M-frame launch ran 2-3.6 %
faster with 25-200 one-wave kernels beside it, tools/lab/fence_cost.py.) Interleaved: plain | a ping every k-th launch, k = 1, 4."""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--rounds", type=int, default=10); ap.add_argument("--reps", type=int, default=400)
a = ap.parse_args()
dev = torch.device("cuda", 0)
A = torch.cuda.Stream(priority=0); B = torch.cuda.Stream(priority=-1); torch.cuda.set_stream(A)
ctx = Context(0); ctx.use_torch_stream(A)
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
out = torch.empty((a.frames, 13), dtype=torch.float32, device=dev)
one = torch.zeros(1, device=dev)
def run(k, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(A)
    for i in range(reps):
        ctx.mfcc_t(bufs[i % 3], a.frames, 1024, _lib.MFCC_B, 13, out=out)
        if k and i % k == 0:
            with torch.cuda.stream(B): one.add_(1.0)
    e1.record(A); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for _ in range(8): run(0, 400)
KS = [0, 1, 4]; t = {k: [] for k in KS}
for r in range(a.rounds):
    for k in (KS if r % 2 == 0 else KS[::-1]):
        run(k, 40); t[k].append(run(k, a.reps))
base = statistics.median(t[0])
for k in KS:
    print("ping every %s launch: median %.2f us  min %.2f us  %+.2f %% vs plain" % (k if k else "no", statistics.median(t[k]), min(t[k]), (base / statistics.median(t[k]) - 1) * 100))
