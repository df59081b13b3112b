#!/usr/bin/env python3
"""Which PAIRS of HIP streams let two 65 536-frame MFCC launches overlap profitably? Streams of several kinds are created in one process
(plain normal priority, high priority, CU-masked with every CU enabled = a hardware queue of its own) and every pair is timed against
the serial sequence, interleaved. usage (GPU box): tools/lab/queue_pairs.py [--reps N] [--rounds R]"""
import argparse, ctypes, itertools, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--reps", type=int, default=200); ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--kinds", default="n,n,n,n,h,h,m,m,m")
ap.add_argument("--tuple", type=int, default=2, help="2: all pairs; 3: all triples (launches rotate over three streams)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(); torch.cuda.set_stream(main)
ctx = Context(0); ctx.use_torch_stream(main)
L = ctx._L
hip = ctypes.CDLL("libamdhip64.so")
streams = []
for k in a.kinds.split(","):
    h = ctypes.c_void_p()
    if k == "n": r = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, 0)
    elif k == "h": r = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, -1)
    else:
        mask = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
        r = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, mask)
    assert r == 0, (k, r)
    streams.append((k + str(sum(1 for x in streams if x[0][0] == k)), h, torch.cuda.ExternalStream(h.value)))
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
outs = [torch.zeros((a.frames, 13), dtype=torch.float32, device=dev) for _ in range(2)]
def call(b, o):
    args = (ctx._h, ctypes.c_void_p(b.data_ptr()), ctypes.c_int64(a.frames), ctypes.c_int64(1024), ctypes.c_int(_lib.MFCC_B), ctypes.c_int(13), ctypes.c_void_p(o.data_ptr()), None, ctypes.c_float(1.0))
    fn = L.edison_mfcc_batch_dev
    return lambda: fn(*args)
T = [call(bufs[i % 3], outs[i % 2]) for i in range(6)]
def setter(h):
    args = (ctx._h, h); fn = L.edison_set_stream
    return lambda: fn(*args)
SETMAIN = setter(ctypes.c_void_p(main.cuda_stream))
def run(pair, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    if pair is None:
        for i in range(reps): T[i % 6]()
    else:
        ss = [setter(streams[p][1]) for p in pair]; st = [streams[p][2] for p in pair]
        for s_ in st: s_.wait_event(e0)
        for i in range(reps): ss[i % len(ss)](); T[i % 6]()
        for s_ in st:
            d = torch.cuda.Event(); d.record(s_); main.wait_event(d)
        SETMAIN()
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for _ in range(10): run(None, 400)
pairs = [None] + list(itertools.combinations(range(len(streams)), a.tuple))
t = {p: [] for p in pairs}
for r in range(a.rounds):
    for p in (pairs if r % 2 == 0 else pairs[::-1]):
        run(p, 40); t[p].append(run(p, a.reps))
base = statistics.median(t[None])
print("serial: %.2f us" % base)
for p in pairs[1:]:
    med = statistics.median(t[p])
    print("%s: %.2f us  %+5.1f %%" % (" + ".join("%-3s" % streams[q][0] for q in p), med, (base / med - 1) * 100))
