#!/usr/bin/env python3
"""Settled time of the serial 65 536-frame launch sequence in THIS process (for A/B over environment variables that only a fresh process
can take, e.g. HIP_FORCE_DEV_KERNARG): prints one line."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(); torch.cuda.set_stream(main)
ctx = Context(0); ctx.use_torch_stream(main)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 65536
bufs = [(torch.randn((N, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
out = torch.zeros((N, 13), dtype=torch.float32, device=dev)
L = ctx._L
calls = []
for b in bufs:
    args = (ctx._h, ctypes.c_void_p(b.data_ptr()), ctypes.c_int64(N), ctypes.c_int64(1024), ctypes.c_int(_lib.MFCC_B), ctypes.c_int(13), ctypes.c_void_p(out.data_ptr()), None, ctypes.c_float(1.0))
    calls.append(lambda a_=args: L.edison_mfcc_batch_dev(*a_))
for i in range(6000): calls[i % 3]()
torch.cuda.synchronize()
ts = []
for r in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for i in range(500): calls[i % 3]()
    e1.record(main); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 500 * 1e3)
print("%s: median %.2f us  min %.2f us" % (os.environ.get("TAG", ""), statistics.median(ts), min(ts)))
