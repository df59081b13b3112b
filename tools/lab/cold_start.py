#!/usr/bin/env python3
"""What a W 5 + K 20 run of the 65 536-frame MFCC step measures, by what the board did just before (lab):
  gen    : the synthetic batches are generated on the GPU (torch RNG kernels, ~tens of ms of full load) right in front of it
  idle   : the board was idle for 2 s
  copy   : idle for 2 s, then the three batches are copied from the host (what a host-generated input does)
Per scenario: mean kernel time of launches 6..25 (the K 20) and of launches 1..60 in blocks of 10, by HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from edison_amd import _lib
from edison_amd.context import Context
import bench
dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.use_torch_stream(st)
nf = 65536
out = torch.empty((nf, 13), dtype=torch.float32, device=dev)
bufs = [bench.synth_frames(nf, 20 + 1000 * r, dev) for r in range(3)]
host = [b.cpu().pin_memory() for b in bufs]
torch.cuda.synchronize()

def run(tag):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for i in range(60):
        ctx.mfcc_t(bufs[i % 3], nf, 1024, _lib.MFCC_B, 13, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(60)]
    print("%-5s K20 (launches 6..25) %.1f us | blocks of 10: %s" % (tag, np.mean(d[5:25]), " ".join("%.1f" % np.mean(d[k:k + 10]) for k in range(0, 60, 10))), flush=True)

for rep in range(3):
    time.sleep(2.0)
    bufs = [bench.synth_frames(nf, 20 + 1000 * r, dev) for r in range(3)]   # not synchronised: the step queues right behind it, as in bench.py
    run("gen")
    time.sleep(2.0)
    run("idle")
    time.sleep(2.0)
    for b, h in zip(bufs, host): b.copy_(h, non_blocking=True)
    run("copy")
