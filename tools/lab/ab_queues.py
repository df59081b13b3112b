#!/usr/bin/env python3
"""Interleaved A/B of the 65 536-frame MFCC launch over 1, 2, 3 HIP queues, per library variant, in ONE process.

Independent batches (own input, own output) alternate over q HIP streams of one context (edison_set_stream per call); the timed
region is forked from / joined into one main stream with events, so the figure is wall time per batch with q launches in flight.
usage (GPU box): tools/lab/ab_queues.py [--frames N] [--rounds R] [--reps K] [--queues 1,2,3] name1 name2 ... ('prod' = product library)
"""
import argparse, ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--queues", default="1,2,3")
ap.add_argument("--variant", type=int, default=_lib.MFCC_B)
ap.add_argument("--trace", action="store_true", help="under rocprofv3: one block of reps per (variant, queues), 5 ms pauses between blocks, no timing table")
ap.add_argument("--prios", default="0,-1", help="priorities of the side streams, cycled")
a = ap.parse_args()
QS = [int(x) for x in a.queues.split(",")]
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(); torch.cuda.set_stream(main)
# Streams of ONE priority share HIP's pool of hardware queues (two torch streams were seen on the same one: launches then serialise);
# pools are per priority, so streams of different priorities are on different hardware queues by construction.
print("stream priority range", torch.cuda.Stream.priority_range(), flush=True)
PRIOS = [int(x) for x in a.prios.split(",")]
side = [torch.cuda.Stream(priority=PRIOS[i % len(PRIOS)]) for i in range(max(QS))]
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]

class V:
    def __init__(self, name):
        p = _lib.LIB_PATH if name == "prod" else os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % name)
        self.name, self.L = name, ctypes.CDLL(p)
        for n in ("edison_init", "edison_set_stream", "edison_mfcc_batch_dev", "edison_last_error"):
            fn = getattr(self.L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
        self.h = ctypes.c_void_p()
        assert self.L.edison_init(0, ctypes.byref(self.h)) == 0
        self.outs = [torch.zeros((a.frames, 13), dtype=torch.float32, device=dev) for _ in range(max(QS + [2]))]
        self.t = {q: [] for q in QS}
        self.cache = {}
    def anyorder(self, on):
        if hasattr(self.L, "ed_lab_set_launch_flags"): self.L.ed_lab_set_launch_flags(1 if on else 0)
        else: assert not on, "queues 0 (any-order launches in one queue) needs a lab library"
    def launch(self, i, stream, out):
        # prepared ctypes arguments: the host must stay well below the ~45 us a batch takes on the GPU, or it -- not the queues -- is
        # what the two-queue figure measures (the first version of this tool converted its arguments per call and was host-bound)
        key = (i % 3, id(stream), id(out))
        c = self.cache.get(key)
        if c is None:
            c = self.cache[key] = ((self.h, ctypes.c_void_p(stream.cuda_stream)),
                                   (self.h, ctypes.c_void_p(bufs[i % 3].data_ptr()), ctypes.c_int64(a.frames), ctypes.c_int64(1024), ctypes.c_int(a.variant),
                                    ctypes.c_int(13), ctypes.c_void_p(out.data_ptr()), None, ctypes.c_float(1.0)))
        self.L.edison_set_stream(*c[0])
        r = self.L.edison_mfcc_batch_dev(*c[1])
        assert r == 0, (self.name, r, self.L.edison_last_error(self.h))
    def run(self, q, reps, timed):
        """reps batches over q queues, forked from and joined into `main`; returns us per batch when timed"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if q == 1:
            for i in range(reps): self.launch(i, main, self.outs[0])
        elif q == 0:   # ONE queue, launches without the barrier bit (hipExtAnyOrderLaunch) behind a first ordered one
            self.launch(0, main, self.outs[0])
            self.anyorder(True)
            for i in range(1, reps): self.launch(i, main, self.outs[i % 2])
            self.anyorder(False)
        else:
            for s in side[:q]: s.wait_event(e0)
            for i in range(reps): self.launch(i, side[i % q], self.outs[i % q])
            for s in side[:q]:
                d = torch.cuda.Event(); d.record(s); main.wait_event(d)
        e1.record(main)
        if timed:
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

_lib._share_torch_hip_runtime()
vs = [V(n) for n in a.names]
# outputs: every variant, serial and pipelined, against the first variant's serial result of batch 0 / 1 / 2
ref = []
for b in range(3):
    vs[0].launch(b, main, vs[0].outs[0]); torch.cuda.synchronize(); ref.append(vs[0].outs[0].clone())
for v in vs:
    for q in QS:
        for o in v.outs: o.zero_()
        v.run(q, 3 * max(q, 1), False); torch.cuda.synchronize()
        # batch i went to queue i % q with input i % 3: the last batch on queue j is i = 3q - q + j
        ok = all(torch.equal(v.outs[j], ref[(3 * q - q + j) % 3]) for j in range(q)) if q else "n/a"
        print("%-10s q=%d outputs bit-identical to %s serial: %s" % (v.name, q, vs[0].name, ok), flush=True)
for i in range(3000): vs[0].launch(i, main, vs[0].outs[0])
torch.cuda.synchronize()
combos = [(v, q) for v in vs for q in QS]
if a.trace:
    import time
    for v, q in combos:
        torch.cuda.synchronize(); time.sleep(0.005)
        t = v.run(q, a.reps, True)
        print("block %s q=%d: %.2f us per batch" % (v.name, q, t), flush=True)
    sys.exit(0)
for r in range(a.rounds):
    for v, q in (combos if r % 2 == 0 else combos[::-1]):
        v.run(q, 60, False)
        v.t[q].append(v.run(q, a.reps, True))
base = statistics.median(vs[0].t[QS[0]])
for v, q in combos:
    med, mn = statistics.median(v.t[q]), min(v.t[q])
    print("%-10s queues %d  median %7.2f us  min %7.2f us  %7.1f Mframes/s  %.4f of 8 TB/s   %+5.1f%% vs %s q=%d" % (
        v.name, q, med, mn, a.frames / med, 2100 * a.frames / med / 1e6 / 8, (base / med - 1) * 100, vs[0].name, QS[0]), flush=True)
