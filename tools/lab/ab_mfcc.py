#!/usr/bin/env python3
"""Interleaved A/B timing of MFCC kernel variants in ONE process (guide rule 24): every variant library is loaded
side by side, rounds alternate between them, medians and minima are reported, and each variant's output is compared
with the first one's.  usage (GPU box): tools/lab/ab_mfcc.py [--frames N] [--rounds R] [--reps K] name1 name2 ...
('prod' = the product library)."""
import argparse, ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--variant", type=int, default=_lib.MFCC_B)
ap.add_argument("--q15", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]

class V:
    def __init__(self, name):
        p = _lib.LIB_PATH if name == "prod" else os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % name)
        self.name, self.L = name, ctypes.CDLL(p)
        for n in ("edison_init", "edison_set_stream", "edison_mfcc_batch_dev", "edison_mfcc_q15_batch_dev", "edison_last_error"):
            fn = getattr(self.L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
        self.h = ctypes.c_void_p()
        assert self.L.edison_init(0, ctypes.byref(self.h)) == 0
        assert self.L.edison_set_stream(self.h, ctypes.c_void_p(st.cuda_stream)) == 0
        self.out = torch.zeros((a.frames, 13), dtype=torch.int16 if a.q15 else torch.float32, device=dev)
        self.t = []
    def launch(self, i):
        if a.q15:
            r = self.L.edison_mfcc_q15_batch_dev(self.h, bufs[i % 3].data_ptr(), a.frames, 1024, 13, self.out.data_ptr(), None)
        else:
            r = self.L.edison_mfcc_batch_dev(self.h, bufs[i % 3].data_ptr(), a.frames, 1024, a.variant, 13, self.out.data_ptr(), None, 1.0)
        assert r == 0, (self.name, r, self.L.edison_last_error(self.h))

_lib._share_torch_hip_runtime()
vs = [V(n) for n in a.names]
for v in vs:
    for i in range(3): v.launch(0)
torch.cuda.synchronize()
ref = vs[0].out.clone()
for v in vs[1:]:
    d = (v.out.double() - ref.double()).abs().max().item()
    print("%-16s max |out - %s| = %g" % (v.name, vs[0].name, d))
# settle the clocks on the first variant, then alternate
for i in range(3000): vs[0].launch(i)
for r in range(a.rounds):
    for v in (vs if r % 2 == 0 else vs[::-1]):
        for i in range(50): v.launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.reps): v.launch(i)
        e1.record(); torch.cuda.synchronize()
        v.t.append(e0.elapsed_time(e1) / a.reps * 1e3)
bytes_per = 2074 if a.q15 else 2100
base = statistics.median(vs[0].t)
for v in vs:
    med, mn = statistics.median(v.t), min(v.t)
    print("%-16s median %7.2f us  min %7.2f us  %6.1f Mframes/s  %.3f of 8 TB/s   %+5.1f%% vs %s" % (
        v.name, med, mn, a.frames / med, bytes_per * a.frames / med / 1e6 / 8, (base / med - 1) * 100, vs[0].name))
